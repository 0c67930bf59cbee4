"""Builds tests/emul/_build/emul_runner: the UNMODIFIED tts_amd/csrc sources compiled for the CPU
with g++ -fsanitize=address,undefined over the HIP stand-in of hip_emul.h.  TEST INFRASTRUCTURE
ONLY (GPU AddressSanitizer is not available on the MI355X pool); the product never loads it.

The only textual changes made to the kernel source on its way to g++ are the two constructs that
have no host spelling: the dynamic-LDS declaration, the "v" (VGPR) inline-asm constraint and the
one-instruction SDWA byte add (rewritten as the C expression it computes)."""
from __future__ import annotations

import os
import re
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "tts_amd", "csrc")
OUT = os.path.join(HERE, "_build")
BIN = os.path.join(OUT, "emul_runner")

_DYN_LDS = re.compile(r"extern\s+__shared__\s+(?:__align__\(\d+\)\s+)?(\w+)\s+(\w+)\[\];")


_SDWA_ADD = re.compile(r'asm\("v_add_u32_sdwa %0, %1, %2 [^"]*src0_sel:BYTE_(\d)[^"]*" : "=v"\((\w+)\) : "s"\((\w+)\), "v"\((\w+)\)\);')


def _translate(src: str) -> str:
    src = _DYN_LDS.sub(r"\1 *\2 = reinterpret_cast<\1 *>(emul::dyn_lds());", src)
    # v_add_u32_sdwa with a byte select on src0: dst = src1 + byte K of src0
    src = _SDWA_ADD.sub(r"\2 = \4 + ((\3 >> (8 * \1)) & 0xFFu);", src)
    # a line that ends in "// EMUL: <statement>" is replaced by that statement
    src = re.sub(r"^[^\n]*// EMUL: ([^\n]*)$", r"\1", src, flags=re.M)
    return src.replace('"+v"', '"+r"')


def build(force: bool = False) -> str:
    srcs = [os.path.join(CSRC, f) for f in ("fz_kernels.hip", "fz_host.cpp", "fz_common.h")]
    srcs += [os.path.join(HERE, f) for f in ("hip_emul.h", "hip_emul.cpp", "emul_runner.cpp", "build_emul.py")]
    srcs.append(os.path.join(ROOT, "include", "fuzzyhip.h"))
    if not force and os.path.exists(BIN) and all(os.path.getmtime(BIN) >= os.path.getmtime(s) for s in srcs):
        return BIN
    os.makedirs(OUT, exist_ok=True)
    gen = os.path.join(OUT, "fz_kernels_emul.cpp")
    with open(os.path.join(CSRC, "fz_kernels.hip")) as f:
        text = _translate(f.read())
    with open(gen, "w") as f:
        f.write(text)
    # -g1: line tables for the sanitizers' stack traces (full -g doubles the kernel file's compile time)
    flags = ["-std=c++20", "-O1", "-g1", "-fsanitize=address,undefined", "-fno-omit-frame-pointer",
             "-ffp-contract=off", "-DFZ_EXACT_ALLOC", "-Wno-unknown-pragmas", "-Wno-attributes",
             "-I" + HERE, "-I" + CSRC, "-I" + os.path.join(ROOT, "include"), "-pthread"]
    units = [gen, os.path.join(CSRC, "fz_host.cpp"), os.path.join(HERE, "hip_emul.cpp"), os.path.join(HERE, "emul_runner.cpp")]
    objs = [os.path.join(OUT, os.path.splitext(os.path.basename(u))[0] + ".o") for u in units]
    procs = [subprocess.Popen(["g++", *flags, "-c", u, "-o", o]) for u, o in zip(units, objs)]   # the four units side by side
    if any(p.wait() != 0 for p in procs):
        raise RuntimeError("emulated build failed")
    subprocess.check_call(["g++", *flags, *objs, "-o", BIN])
    return BIN


if __name__ == "__main__":
    print(build(force=True))
