"""The scene front-end's TOML reader (rt_amd/host/toml_subset.hpp) against an independent parser (tomli).

The reference reads scenes with toml++ (src/scene.cpp:3, 531-560); the mirror has its own reader for the subset scene
files use.  These tests generate documents in varied surface syntax (dotted keys, inline tables, arrays of tables,
multi-line arrays with comments, every string flavour, every integer base, floats incl. inf/nan) and require the
same tree from both parsers; malformed documents must be rejected by both."""
import ctypes as C
import json
import math
import random

import pytest
import tomli

from rt_amd import capi


def to_tree(text: str):
    lib = capi.host_lib()
    buf = C.create_string_buffer(1 << 20)
    n = lib.rt_host_toml_to_json(text.encode("utf-8"), buf, len(buf))
    if n < 0:
        raise ValueError(lib.rt_host_last_error().decode())
    assert n < len(buf)
    return json.loads(buf.value.decode("utf-8"))


def normalise(tree):
    """tomli's tree in the JSON dump's conventions (non-finite floats as strings)."""
    if isinstance(tree, dict):
        return {k: normalise(v) for k, v in tree.items()}
    if isinstance(tree, list):
        return [normalise(v) for v in tree]
    if isinstance(tree, float):
        if math.isnan(tree):
            return "nan"
        if math.isinf(tree):
            return "-inf" if tree < 0 else "inf"
    return tree


def same(a, b) -> bool:
    if type(a) is not type(b):
        return False
    if isinstance(a, dict):
        return list(a.keys()) == list(b.keys()) and all(same(a[k], b[k]) for k in a)
    if isinstance(a, list):
        return len(a) == len(b) and all(same(x, y) for x, y in zip(a, b))
    return a == b


# ---------------------------------------------------------------------------------------------------------------------
# a small document generator
# ---------------------------------------------------------------------------------------------------------------------
BARE = "abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789_-"


class Emitter:
    def __init__(self, rng: random.Random):
        self.rng = rng

    def key(self) -> str:
        r = self.rng
        name = "".join(r.choice(BARE) for _ in range(r.randint(1, 8)))
        style = r.random()
        if style < 0.75:
            return name
        if style < 0.9:
            return '"' + name + ' k"'
        return "'" + name + ".k'"

    def string(self) -> str:
        r = self.rng
        body = "".join(r.choice("abc xyz_-#=[]{},.09") for _ in range(r.randint(0, 12)))
        style = r.random()
        if style < 0.4:
            return '"' + body + r.choice(["", "\\t", "\\n", '\\"', "\\\\", "\\u00e9", "\\U0001F600"]) + '"'
        if style < 0.7:
            return "'" + body + r.choice(["", "\\", "\\n"]) + "'"
        if style < 0.85:
            return '"""\n' + body + "\n second \\\n   joined" + '"""'
        return "'''\n" + body + "\nraw \\n'''"

    def integer(self) -> str:
        r = self.rng
        v = r.choice([0, 1, 7, 42, 255, 1000, 65535, 2**31, 2**53, 2**62])
        style = r.random()
        if style < 0.5:
            return r.choice(["", "+", "-"]) + str(v)
        if style < 0.6 and v >= 1000:
            s = str(v)
            return s[:-3] + "_" + s[-3:]
        if style < 0.75:
            return hex(v)
        if style < 0.85:
            return "0x" + format(v, "X")
        if style < 0.93:
            return oct(v)
        return bin(v)

    def floating(self) -> str:
        r = self.rng
        style = r.random()
        if style < 0.1:
            return r.choice(["inf", "+inf", "-inf", "nan", "+nan", "-nan"])
        mant = r.choice(["0.5", "1.0", "3.25", "1000.001", "0.05", "6.02", "9_000.5"])
        exp = r.choice(["", "", "e3", "E-2", "e+10", "e-30"])
        if r.random() < 0.2:
            mant = mant.split(".")[0]
            exp = exp or "e0"
        return r.choice(["", "+", "-"]) + mant + exp

    def scalar(self) -> str:
        r = self.rng.random()
        if r < 0.3:
            return self.string()
        if r < 0.6:
            return self.integer()
        if r < 0.9:
            return self.floating()
        return self.rng.choice(["true", "false"])

    def value(self, depth: int) -> str:
        r = self.rng.random()
        if depth < 2 and r < 0.2:
            return self.array(depth + 1)
        if depth < 2 and r < 0.3:
            return self.inline_table(depth + 1)
        return self.scalar()

    def array(self, depth: int) -> str:
        r = self.rng
        items = [self.value(depth) for _ in range(r.randint(0, 5))]
        if r.random() < 0.4:
            body = ",\n    ".join(items)
            return "[\n    " + body + (", # trailing\n" if items else "# empty\n") + "]"
        return "[" + ", ".join(items) + (" " if r.random() < 0.5 else "") + "]"

    def inline_table(self, depth: int) -> str:
        r = self.rng
        keys = self.unique_keys(r.randint(0, 4))
        return "{ " + ", ".join(f"{k} = {self.value(depth)}" for k in keys) + " }" if keys else "{}"

    def unique_keys(self, n: int) -> list:
        seen, out = set(), []
        while len(out) < n:
            k = self.key()
            plain = k.strip("\"'")
            if plain not in seen:
                seen.add(plain)
                out.append(k)
        return out

    def body(self, depth: int = 0) -> list:
        r = self.rng
        lines = []
        for k in self.unique_keys(r.randint(0, 5)):
            if r.random() < 0.15:
                sub = self.unique_keys(2)
                lines.append(f"{k}.{sub[0]} = {self.value(depth)}")
                lines.append(f"{k} . {sub[1]} = {self.value(depth)}   # dotted")
            else:
                pad = " " * r.randint(0, 3)
                lines.append(f"{k}{pad}={pad}{self.value(depth)}" + ("  # note" if r.random() < 0.2 else ""))
        return lines

    def document(self) -> str:
        r = self.rng
        lines = ["# generated"] + self.body()
        for t in self.unique_keys(r.randint(0, 4)):
            kind = r.random()
            if kind < 0.5:
                lines += ["", f"[{t}]"] + self.body()
                if r.random() < 0.4:
                    (child,) = self.unique_keys(1)
                    lines += [f"[{t}.{child}]"] + self.body()
            else:
                for _ in range(r.randint(1, 3)):
                    lines += ["", f"[[{t}]]"] + self.body()
                    if r.random() < 0.3:
                        lines += [f"  [{t}.nested]"] + self.body()
        return "\n".join(lines) + ("\n" if r.random() < 0.8 else "")


@pytest.mark.parametrize("seed", range(300))
def test_generated_documents_parse_to_the_same_tree(seed):
    text = Emitter(random.Random(seed)).document()
    expected = normalise(tomli.loads(text))
    got = to_tree(text)
    assert same(got, expected), f"seed {seed}\n{text}\n---\n{got}\n---\n{expected}"


def test_the_shipped_scenes_parse_to_the_same_tree():
    import pathlib

    for path in sorted((pathlib.Path(__file__).resolve().parents[1] / "scenes").glob("*.toml")):
        text = path.read_text()
        assert same(to_tree(text), normalise(tomli.loads(text))), path.name


MALFORMED = [
    "a = ",
    "a = 1\na = 2",
    "[t]\n[t]",
    "a = [1, 2",
    'a = "unterminated',
    "a = 'unterminated",
    "a = -0x10",
    "a = 01",
    "a = 1__0",
    "a = _1",
    "a = 1_",
    "a = 1.",
    "a = .5",
    "a = 1e",
    "= 1",
    "a b = 1",
    "[t",
    "[[t]\n",
    "[]",
    "a = 1 b = 2",
    'a = "bad \\q escape"',
    "a = tru",
    "a.b = 1\na = 2",
    "a = 1\na.b = 2",
    "[a]\nb = 1\n[a.b]\n",
    "[[a]]\n[a]\n",
    "a = [1]\n[[a]]\n",
    "a = {}\n[a.b]\n",
]


@pytest.mark.parametrize("text", MALFORMED)
def test_malformed_documents_are_rejected_like_the_independent_parser(text):
    with pytest.raises(tomli.TOMLDecodeError):
        tomli.loads(text)
    with pytest.raises(ValueError):
        to_tree(text)


@pytest.mark.parametrize("text", ["a = {b = 1,}", "a = {b = 1\n}", "a = {\n b = 1 }"])
def test_toml_1_1_inline_table_extensions_are_rejected(text):
    """toml++ as the reference builds it (no TOML_ENABLE_UNRELEASED_FEATURES) reads TOML 1.0: inline tables stay on
    one line and take no trailing comma.  (tomli 2.4 already reads TOML 1.1, so it is no witness here.)"""
    with pytest.raises(ValueError):
        to_tree(text)


def test_a_short_output_buffer_is_truncated_and_the_needed_size_reported():
    lib = capi.host_lib()
    buf = C.create_string_buffer(8)
    n = lib.rt_host_toml_to_json(b'key = "a long enough value"', buf, len(buf))
    assert n == len('{"key":"a long enough value"}')
    assert buf.value == b'{"key":'
