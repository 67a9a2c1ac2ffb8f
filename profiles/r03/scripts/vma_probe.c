#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/prctl.h>
#include <sys/ioctl.h>
#include <fcntl.h>
#include <errno.h>
#include <unistd.h>
struct procmap_query { uint64_t size, query_flags, query_addr, vma_start, vma_end, vma_flags, vma_page_size, vma_offset, inode; uint32_t dev_major, dev_minor, vma_name_size, build_id_size; uint64_t vma_name_addr, build_id_addr; };
#define PROCMAP_QUERY _IOWR('f', 17, struct procmap_query)
int main(){
  size_t len = 1<<20; void* p = mmap(0,len,PROT_READ|PROT_WRITE,MAP_PRIVATE|MAP_ANONYMOUS,-1,0);
  int r = prctl(0x53564d41, 0, (unsigned long)p, len, (unsigned long)"rt_hip_frame_1");
  printf("prctl name: %d errno %d\n", r, errno);
  int fd = open("/proc/self/maps", O_RDONLY);
  char name[128]; struct procmap_query q; memset(&q,0,sizeof q); q.size=sizeof q; q.query_addr=(uint64_t)p; q.vma_name_size=sizeof name; q.vma_name_addr=(uint64_t)name;
  r = ioctl(fd, PROCMAP_QUERY, &q);
  printf("ioctl: %d errno %d start %lx end %lx name '%s' (%u)\n", r, errno, (long)q.vma_start,(long)q.vma_end, r==0?name:"", q.vma_name_size);
  munmap(p,len); void* p2 = mmap(p,len,PROT_READ|PROT_WRITE,MAP_PRIVATE|MAP_ANONYMOUS|MAP_FIXED_NOREPLACE,-1,0);
  memset(&q,0,sizeof q); q.size=sizeof q; q.query_addr=(uint64_t)p; q.vma_name_size=sizeof name; q.vma_name_addr=(uint64_t)name; name[0]=0;
  r = ioctl(fd, PROCMAP_QUERY, &q);
  printf("after remap (%p==%p): ioctl %d errno %d name '%s' (%u)\n", p,p2,r,errno,name,q.vma_name_size);
  return 0; }
