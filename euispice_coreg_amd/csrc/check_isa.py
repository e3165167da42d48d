#!/usr/bin/env python3
"""Build-time check of the hand-issued LDS gathers of k_sweep (kernels.hpp, struct Taps).

The nine (four) `ds_read_b64` of a sample and their `s_waitcnt lgkmcnt(N)` live in two separate inline-asm statements so
that the spline weights -- and, in the software-pipelined interior loop, the previous sample's arithmetic -- run under
the LDS latency (N = 9: the next sample's reads stay in flight; LDS operations return in order).  The compiler does not track memory operations inside
inline asm: were it to place a copy, a spill or any other use of a tap register between the reads and the wait, that
instruction would see stale data.  This script disassembles the built library and fails if any instruction between a
group of hand-issued reads and the wait that follows it names one of the group's destination registers.

usage: check_isa.py libcoreg_hip.so   (needs llvm-objdump / clang-offload-bundler from /opt/rocm/lib/llvm/bin)
"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = os.environ.get("ROCM_LLVM_BIN", "/opt/rocm/lib/llvm/bin")


def disassemble(lib):
    with tempfile.TemporaryDirectory() as d:
        fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "dev.co")
        subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", lib, fat],
                       check=True, capture_output=True)
        subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--type=o", "--unbundle",
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={fat}", f"--output={co}"], check=True,
                       capture_output=True)
        if os.path.getsize(co) == 0:
            raise RuntimeError("no gfx950 code object in " + lib)
        return subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--mcpu=gfx950", co], check=True,
                              capture_output=True, text=True).stdout


def regs(tok):
    """Register numbers named by an operand token such as v12, v[50:51] (VGPRs only)."""
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]", tok):
        out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\bv(\d+)\b", tok):
        out.add(int(m.group(1)))
    return out


def check(text):
    n_groups, bad = 0, []
    kernel = "?"
    lines = text.splitlines()
    i = 0
    while i < len(lines):
        ln = lines[i]
        m = re.match(r"^[0-9a-f]+ <(.+)>:", ln)
        if m:
            kernel = m.group(1)
        ins = ln.strip().split("//")[0].strip()
        if ins.startswith("ds_read_b64") and "k_sweep" in kernel:
            # a hand-issued group: consecutive ds_read_b64
            dst = set()
            j = i
            while j < len(lines) and lines[j].strip().startswith("ds_read_b64"):
                ops = lines[j].strip().split("//")[0].split(None, 1)[1]
                dst |= regs(ops.split(",")[0])
                j += 1
            if j - i in (4, 9):  # Taps<2> / Taps<3>
                n_groups += 1
                k = j
                younger = 0  # LDS reads issued after this group (software-pipelined loop: the next sample's)
                while k < len(lines):
                    t = lines[k].strip().split("//")[0].strip()
                    m2 = re.match(r"s_waitcnt .*lgkmcnt\((\d+)\)", t)
                    if m2 and int(m2.group(1)) <= younger:
                        break  # in-order return: this group's reads have landed
                    if t.startswith("ds_read"):
                        younger += 1
                    if t and not t.startswith(("s_nop", ";")):
                        ops = t.split(None, 1)[1] if " " in t else ""
                        if regs(ops) & dst:
                            bad.append((kernel, t))
                    if t.startswith(("s_endpgm", "s_branch", "s_cbranch")):
                        # a short FORWARD conditional branch (the EXEC-masked accumulation of the previous sample in
                        # the software-pipelined loop) only skips instructions that are checked here anyway
                        m3 = re.match(r"s_cbranch_\w+\s+(\d+)", t)
                        if not (m3 and int(m3.group(1)) < 64):
                            bad.append((kernel, "control flow before the wait: " + t))
                            break
                    k += 1
            i = j
            continue
        i += 1
    return n_groups, bad


def main(lib):
    n, bad = check(disassemble(lib))
    if n == 0:
        raise SystemExit("check_isa: no hand-issued ds_read_b64 group found in k_sweep (disassembly format changed?)")
    if bad:
        for k, t in bad[:20]:
            print("check_isa: tap register touched between the reads and their wait:", k, "|", t, file=sys.stderr)
        raise SystemExit(f"check_isa: {len(bad)} violation(s)")
    print(f"[check_isa] ok: {n} hand-issued LDS read groups, none has its registers touched before the wait")


if __name__ == "__main__":
    main(sys.argv[1])
