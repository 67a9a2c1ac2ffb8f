#!/usr/bin/env python3
"""Writes tools/valu_rate_probe2.hip: issue cost of vector instructions with PHYSICAL registers chosen by hand, so that
register-bank effects are part of the question instead of an accident of the compiler's allocation.  16 instructions per
trip, every destination different from its sources, results feeding instructions two trips later at the earliest.
v32..v47 are the working registers (v32+i), s20..s27 scalar operands."""
from pathlib import Path

R = lambda i: f"v{32 + (i % 16)}"
RP = lambda i: f"v[{32 + 2 * (i % 8)}:{33 + 2 * (i % 8)}]"

def body(fmt, n=16):
    # fmt(i) -> instruction text for slot i
    return "\\n".join(fmt(i) for i in range(n))

FORMS = {
    # name: (instruction for slot i)
    # sources in three DIFFERENT banks (index mod 4 all differ): i+1, i+2, i+3 relative to destination i
    "v_fma_f32 3 banks": lambda i: f"v_fma_f32 {R(i)}, {R(i+1)}, {R(i+2)}, {R(i+3)}",
    # two sources in the SAME bank (i+1 and i+5), third elsewhere
    "v_fma_f32 2 srcs one bank": lambda i: f"v_fma_f32 {R(i)}, {R(i+1)}, {R(i+5)}, {R(i+2)}",
    # all three sources in the same bank
    "v_fma_f32 3 srcs one bank": lambda i: f"v_fma_f32 {R(i)}, {R(i+1)}, {R(i+5)}, {R(i+9)}",
    "v_fma_f32 x,x,y": lambda i: f"v_fma_f32 {R(i)}, {R(i+1)}, {R(i+1)}, {R(i+2)}",
    "v_fma_f32 x,y,y": lambda i: f"v_fma_f32 {R(i)}, {R(i+1)}, {R(i+2)}, {R(i+2)}",
    "v_fmac_f32 2 banks": lambda i: f"v_fmac_f32 {R(i)}, {R(i+1)}, {R(i+2)}",
    "v_fmac_f32 x,x": lambda i: f"v_fmac_f32 {R(i)}, {R(i+1)}, {R(i+1)}",
    "v_fma_f32 v,s,v": lambda i: f"v_fma_f32 {R(i)}, {R(i+1)}, s{20 + i % 8}, {R(i+2)}",
    "v_mul_f32 v,v": lambda i: f"v_mul_f32 {R(i)}, {R(i+1)}, {R(i+2)}",
    "v_mul_f32 same bank": lambda i: f"v_mul_f32 {R(i)}, {R(i+1)}, {R(i+5)}",
    "v_add_f32 v,v": lambda i: f"v_add_f32 {R(i)}, {R(i+1)}, {R(i+2)}",
    "v_sub_f32 s,v": lambda i: f"v_sub_f32 {R(i)}, s{20 + i % 8}, {R(i+1)}",
    "v_mul_f32 s,v": lambda i: f"v_mul_f32 {R(i)}, s{20 + i % 8}, {R(i+1)}",
    "v_max_f32 v,v": lambda i: f"v_max_f32 {R(i)}, {R(i+1)}, {R(i+2)}",
    "v_cmp_lt_f32 vcc": lambda i: f"v_cmp_lt_f32 vcc, {R(i+1)}, {R(i+2)}",
    "v_cmp_lt_f32 s[28:29]": lambda i: f"v_cmp_lt_f32 s[28:29], {R(i+1)}, {R(i+2)}",
    "v_cmp_lt_f32 s,v -> s[28:29]": lambda i: f"v_cmp_lt_f32 s[28:29], s{20 + i % 8}, {R(i+2)}",
    "v_cndmask_b32 vcc": lambda i: f"v_cndmask_b32 {R(i)}, {R(i+1)}, {R(i+2)}, vcc",
    "v_cndmask_b32 s[30:31]": lambda i: f"v_cndmask_b32 {R(i)}, {R(i+1)}, {R(i+2)}, s[30:31]",
    "v_mov_b32": lambda i: f"v_mov_b32 {R(i)}, {R(i+1)}",
    "v_add_u32": lambda i: f"v_add_u32 {R(i)}, {R(i+1)}, {R(i+2)}",
    "v_xor_b32": lambda i: f"v_xor_b32 {R(i)}, {R(i+1)}, {R(i+2)}",
    "v_lshrrev_b32": lambda i: f"v_lshrrev_b32 {R(i)}, 15, {R(i+1)}",
    "v_xor_b32_sdwa": lambda i: f"v_xor_b32_sdwa {R(i)}, {R(i+1)}, {R(i+1)} dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD",
    "v_mul_lo_u32": lambda i: f"v_mul_lo_u32 {R(i)}, {R(i+1)}, {R(i+2)}",
    "v_mul_lo_u32 v,s": lambda i: f"v_mul_lo_u32 {R(i)}, {R(i+1)}, s{20 + i % 8}",
    "v_mad_u64_u32": lambda i: f"v_mad_u64_u32 {RP(i)}, s[28:29], {R(2*i+3)}, {R(2*i+5)}, {RP(i+1)}",
    "v_cvt_f32_u32": lambda i: f"v_cvt_f32_u32 {R(i)}, {R(i+1)}",
    "v_rsq_f32": lambda i: f"v_rsq_f32 {R(i)}, {R(i+1)}",
    "v_sqrt_f32": lambda i: f"v_sqrt_f32 {R(i)}, {R(i+1)}",
    "v_rcp_f32": lambda i: f"v_rcp_f32 {R(i)}, {R(i+1)}",
    "v_pk_fma_f32": lambda i: f"v_pk_fma_f32 {RP(i)}, {RP(i+1)}, {RP(i+2)}, {RP(i+3)}",
    "v_pk_mul_f32": lambda i: f"v_pk_mul_f32 {RP(i)}, {RP(i+1)}, {RP(i+2)}",
    "v_pk_add_f32": lambda i: f"v_pk_add_f32 {RP(i)}, {RP(i+1)}, {RP(i+2)}",
    "v_pk_add_f32 s,v": lambda i: f"v_pk_add_f32 {RP(i)}, s[{20 + 2 * (i % 4)}:{21 + 2 * (i % 4)}], {RP(i+2)} neg_lo:[0,1] neg_hi:[0,1]",
    "v_pk_fma_f32 lo-broadcast": lambda i: f"v_pk_fma_f32 {RP(i)}, {RP(i+1)}, {RP(i+2)}, {RP(i+3)} op_sel_hi:[0,1,1]",
    "v_fma_f64": lambda i: f"v_fma_f64 {RP(i)}, {RP(i+1)}, {RP(i+2)}, {RP(i+3)}",
    "v_mul_f64": lambda i: f"v_mul_f64 {RP(i)}, {RP(i+1)}, {RP(i+2)}",
    # mixes: eight of the instruction in question alternating with eight plain v_fma_f32 (2.2 cycles each on their own)
    "mix: fma + v_sub_f32 s,v": lambda i: (f"v_sub_f32 {R(i)}, s{20 + i % 8}, {R(i+1)}" if i % 2 else f"v_fma_f32 {R(i)}, {R(i+1)}, {R(i+2)}, {R(i+3)}"),
    "mix: fma + v_cmp_lt_f32": lambda i: (f"v_cmp_lt_f32 s[28:29], {R(i+1)}, {R(i+2)}" if i % 2 else f"v_fma_f32 {R(i)}, {R(i+1)}, {R(i+2)}, {R(i+3)}"),
    "mix: fma + v_cndmask_b32": lambda i: (f"v_cndmask_b32 {R(i)}, {R(i+1)}, {R(i+2)}, s[30:31]" if i % 2 else f"v_fma_f32 {R(i)}, {R(i+1)}, {R(i+2)}, {R(i+3)}"),
    "mix: fma + v_max_f32": lambda i: (f"v_max_f32 {R(i)}, {R(i+1)}, {R(i+2)}" if i % 2 else f"v_fma_f32 {R(i)}, {R(i+1)}, {R(i+2)}, {R(i+3)}"),
    "mix: fma + v_mul_lo_u32": lambda i: (f"v_mul_lo_u32 {R(i)}, {R(i+1)}, {R(i+2)}" if i % 2 else f"v_fma_f32 {R(i)}, {R(i+1)}, {R(i+2)}, {R(i+3)}"),
    "mix: fma + v_cvt_f32_u32": lambda i: (f"v_cvt_f32_u32 {R(i)}, {R(i+1)}" if i % 2 else f"v_fma_f32 {R(i)}, {R(i+1)}, {R(i+2)}, {R(i+3)}"),
    "mix: fma + v_xor_b32_sdwa": lambda i: (f"v_xor_b32_sdwa {R(i)}, {R(i+1)}, {R(i+1)} dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD" if i % 2 else f"v_fma_f32 {R(i)}, {R(i+1)}, {R(i+2)}, {R(i+3)}"),
    "mix: fma + v_rsq_f32": lambda i: (f"v_rsq_f32 {R(i)}, {R(i+1)}" if i % 2 else f"v_fma_f32 {R(i)}, {R(i+1)}, {R(i+2)}, {R(i+3)}"),
    "mix: 3 fma + v_rsq_f32": lambda i: (f"v_rsq_f32 {R(i)}, {R(i+1)}" if i % 4 == 3 else f"v_fma_f32 {R(i)}, {R(i+1)}, {R(i+2)}, {R(i+3)}"),
    "mix: 3 fma + v_cmp_lt_f32": lambda i: (f"v_cmp_lt_f32 s[28:29], {R(i+1)}, {R(i+2)}" if i % 4 == 3 else f"v_fma_f32 {R(i)}, {R(i+1)}, {R(i+2)}, {R(i+3)}"),
    "mix: 3 fma + v_cndmask_b32": lambda i: (f"v_cndmask_b32 {R(i)}, {R(i+1)}, {R(i+2)}, s[30:31]" if i % 4 == 3 else f"v_fma_f32 {R(i)}, {R(i+1)}, {R(i+2)}, {R(i+3)}"),
    "mix: 3 fma + v_mul_lo_u32": lambda i: (f"v_mul_lo_u32 {R(i)}, {R(i+1)}, {R(i+2)}" if i % 4 == 3 else f"v_fma_f32 {R(i)}, {R(i+1)}, {R(i+2)}, {R(i+3)}"),
    "mix: fma + v_pk_fma_f32": lambda i: (f"v_pk_fma_f32 {RP(i)}, {RP(i+1)}, {RP(i+2)}, {RP(i+3)}" if i % 2 else f"v_fma_f32 {R(i)}, {R(i+1)}, {R(i+2)}, {R(i+3)}"),
    "mix: fma + v_fma_f64": lambda i: (f"v_fma_f64 {RP(i)}, {RP(i+1)}, {RP(i+2)}, {RP(i+3)}" if i % 2 else f"v_fma_f32 {R(i)}, {R(i+1)}, {R(i+2)}, {R(i+3)}"),
    "mix: fma + s_and_b64 (scalar)": lambda i: (f"s_and_b64 s[28:29], s[30:31], s[20:21]" if i % 2 else f"v_fma_f32 {R(i)}, {R(i+1)}, {R(i+2)}, {R(i+3)}"),
    # the sphere probe of the streamed kernel as it is compiled (12 instructions; v32..v34 = origin, v35..v37 = direction)
    "sphere probe (12 insts)": None,
    "two spheres packed (13 insts)": None,
}

PROBE = [
    "v_sub_f32 v40, s20, v32", "v_sub_f32 v41, s21, v33", "v_mul_f32 v43, v35, v40", "v_mul_f32 v44, v40, v40", "v_sub_f32 v42, s22, v34",
    "v_fmac_f32 v43, v41, v36", "v_fmac_f32 v44, v41, v41", "v_fmac_f32 v43, v42, v37", "v_fmac_f32 v44, v42, v42", "v_fma_f32 v45, -v43, v43, v44",
    "v_sub_f32 v46, s23, v45", "v_cmp_ngt_f32 s[28:29], 0, v46",
]
# two spheres at once: s[20:21] = (cx_A, cx_B), s[22:23] = cy, s[24:25] = cz, s[26:27] = r2; origin/direction components broadcast from single registers
PACKED = [
    "v_pk_add_f32 v[40:41], s[20:21], v[32:33] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]",
    "v_pk_add_f32 v[42:43], s[22:23], v[32:33] op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]",
    "v_pk_add_f32 v[44:45], s[24:25], v[34:35] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]",
    "v_pk_mul_f32 v[46:47], v[40:41], v[36:37] op_sel_hi:[1,0]",
    "v_pk_mul_f32 v[48:49], v[40:41], v[40:41]",
    "v_pk_fma_f32 v[46:47], v[42:43], v[36:37], v[46:47] op_sel:[0,1,0] op_sel_hi:[1,1,1]",
    "v_pk_fma_f32 v[48:49], v[42:43], v[42:43], v[48:49]",
    "v_pk_fma_f32 v[46:47], v[44:45], v[38:39], v[46:47] op_sel_hi:[1,0,1]",
    "v_pk_fma_f32 v[48:49], v[44:45], v[44:45], v[48:49]",
    "v_pk_fma_f32 v[50:51], v[46:47], v[46:47], v[48:49] neg_lo:[1,0,0] neg_hi:[1,0,0]",
    "v_pk_add_f32 v[52:53], s[26:27], v[50:51] neg_lo:[0,1] neg_hi:[0,1]",
    "v_cmp_ngt_f32 s[28:29], 0, v52", "v_cmp_ngt_f32 s[30:31], 0, v53",
]

def text_of(name, f):
    if name.startswith("sphere probe"):
        return "\\n".join(PROBE), len(PROBE)
    if name.startswith("two spheres"):
        return "\\n".join(PACKED), len(PACKED)
    return body(f), 16

out = ['// GENERATED by tools/gen_valu_rate_probe2.py — do not edit.  hipcc --offload-arch=gfx950 -O3 -o tools/valu_rate_probe2 tools/valu_rate_probe2.hip',
       '#include <hip/hip_runtime.h>', '#include <cstdio>', '#include <cstdlib>', 'constexpr int trips = 8192;']
clob = ", ".join(f'"v{r}"' for r in range(32, 56)) + ', "s20","s21","s22","s23","s24","s25","s26","s27","s28","s29","s30","s31","vcc"'
names = []
for k, (name, f) in enumerate(FORMS.items()):
    text, n = text_of(name, f)
    names.append((name, n))
    init = "\\n".join([f"v_mov_b32 v{32+i}, %0" if i % 3 else f"v_mov_b32 v{32+i}, %1" for i in range(24)] + [f"s_mov_b32 s{20+i}, 0x3f8{i}0347" for i in range(8)] + ["s_mov_b32 s30, 0x55555555", "s_mov_b32 s31, 0x33333333", "v_cmp_lt_u32 vcc, %0, %1"])
    out.append(f'''__global__ void __launch_bounds__(256) probe{k}(uint32_t* out, uint32_t seed)
{{
	const float x = 1.0f + 1e-6f * threadIdx.x, y = 0.999f + 1e-7f * seed;
	asm volatile("{init}" : : "v"(x), "v"(y) : {clob});
	for (int i = 0; i < trips; i++)
		asm volatile("{text}" : : : {clob});
	uint32_t r;
	asm volatile("v_xor_b32 %0, v32, v33\\nv_xor_b32 %0, %0, v40\\nv_xor_b32 %0, %0, v46" : "=v"(r) : : {clob});
	out[blockIdx.x * 256u + threadIdx.x] = r;
}}''')
out.append('int main()\n{\n\thipDeviceProp_t prop;\n\tif (hipGetDeviceProperties(&prop, 0) != hipSuccess) return 1;\n\tconst int blocks = prop.multiProcessorCount * 8;\n\tuint32_t* out = nullptr;\n\tif (hipMalloc(&out, size_t(blocks) * 1024) != hipSuccess) return 1;\n\thipEvent_t t0, t1;\n\thipEventCreate(&t0);\n\thipEventCreate(&t1);')
out.append('\tstruct { const char* name; void (*kernel)(uint32_t*, uint32_t); int insts; } probes[] = {')
for k, (name, n) in enumerate(names):
    out.append(f'\t\t{{ "{name}", probe{k}, {n} }},')
out.append('\t};\n\tconstexpr int n = sizeof(probes) / sizeof(probes[0]);\n\tdouble best[n];\n\tfor (double& b : best) b = 1e30;')
out.append('\tfor (int w = 0; w < 30; w++) hipLaunchKernelGGL(probes[0].kernel, dim3(blocks), dim3(256), 0, 0, out, 1u);')
out.append('''	for (int pass = 0; pass < 4; pass++)
		for (int i = 0; i < n; i++)
		{
			hipLaunchKernelGGL(probes[i].kernel, dim3(blocks), dim3(256), 0, 0, out, 1u);
			hipEventRecord(t0);
			for (int r = 0; r < 3; r++) hipLaunchKernelGGL(probes[i].kernel, dim3(blocks), dim3(256), 0, 0, out, 1u);
			hipEventRecord(t1);
			hipEventSynchronize(t1);
			float ms = 0;
			hipEventElapsedTime(&ms, t0, t1);
			if (ms / 3 < best[i]) best[i] = ms / 3;
		}
	for (int i = 0; i < n; i++)
	{
		const double wave_insts_per_simd = 8.0 * trips * probes[i].insts; // 8 waves per SIMD
		std::printf("%-32s %8.4f ms  %5.2f cycles per wave-instruction (%d per trip; %6.1f cycles per trip) if the clock is 2.4 GHz\\n", probes[i].name, best[i], best[i] * 1e-3 * 2.4e9 / wave_insts_per_simd, probes[i].insts, best[i] * 1e-3 * 2.4e9 / (8.0 * trips));
	}
	return 0;
}''')
Path(__file__).with_name("valu_rate_probe2.hip").write_text("\n".join(out) + "\n")
