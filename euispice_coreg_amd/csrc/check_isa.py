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
        # the device targets the library was actually built for (not assumed): every one of them is disassembled
        listing = subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--type=o", "--list", f"--input={fat}"],
                                 check=True, capture_output=True, text=True).stdout.split()
        targets = [t for t in listing if t.startswith("hip") and "amdgcn" in t]
        if not targets:
            raise RuntimeError("no AMD GPU code object in " + lib)
        text = []
        for t in targets:
            subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--type=o", "--unbundle", f"--targets={t}",
                            f"--input={fat}", f"--output={co}"], check=True, capture_output=True)
            if os.path.getsize(co) == 0:
                raise RuntimeError(f"empty code object for {t} in {lib}")
            arch = t.split("-")[-1].split(":")[0]
            text.append(subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", f"--mcpu={arch}", co], check=True,
                                       capture_output=True, text=True).stdout)
        return "\n".join(text)


def kernel_metadata(lib):
    """[(kernel name, static LDS bytes, VGPRs, spilled VGPRs + scratch bytes)] from the code objects' metadata notes."""
    out = []
    with tempfile.TemporaryDirectory() as d:
        fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "dev.co")
        subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", lib, fat],
                       check=True, capture_output=True)
        listing = subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--type=o", "--list", f"--input={fat}"],
                                 check=True, capture_output=True, text=True).stdout.split()
        for t in [t for t in listing if t.startswith("hip") and "amdgcn" in t]:
            subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--type=o", "--unbundle", f"--targets={t}",
                            f"--input={fat}", f"--output={co}"], check=True, capture_output=True)
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], check=True, capture_output=True,
                                   text=True).stdout
            cur = {}
            for ln in notes.splitlines():
                m = re.match(r"\s*-?\s*\.(\w+):\s*(.+?)\s*$", ln)
                if not m:
                    continue
                key, val = m.group(1), m.group(2).strip("'\"")
                if key == "agpr_count" and cur.get("name"):  # (first key of a kernel entry: flush the previous one)
                    out.append(cur)
                    cur = {}
                cur[key] = val
            if cur.get("name"):
                out.append(cur)
    return [(k.get("name", "?"), int(k.get("group_segment_fixed_size", 0)), int(k.get("vgpr_count", 0)),
             int(k.get("vgpr_spill_count", 0)) + int(k.get("private_segment_fixed_size", 0))) for k in out if "name" in k]


# dynamic LDS the host asks for at most (host_state.hpp: opt_lds_bytes) and what a gfx950 workgroup can have
MAX_DYNAMIC_LDS = 159 * 1024
LDS_PER_WORKGROUP = 160 * 1024


def check_static_lds(lib):
    """k_sweep's static LDS + the largest dynamic window must fit a workgroup's LDS (hipFuncSetAttribute fails at run
    time otherwise, on the GPU box only)."""
    meta = [m for m in kernel_metadata(lib) if "k_sweep" in m[0]]
    if not meta:
        raise SystemExit("check_isa: no k_sweep kernel in the metadata notes (format changed?)")
    worst = max(meta, key=lambda m: m[1])
    if worst[1] + MAX_DYNAMIC_LDS > LDS_PER_WORKGROUP:
        raise SystemExit(f"check_isa: {worst[0]} has {worst[1]} B of static LDS: with the {MAX_DYNAMIC_LDS} B dynamic "
                         f"window that exceeds the {LDS_PER_WORKGROUP} B of a workgroup")
    return len(meta), worst[1], max(m[2] for m in meta), sum(1 for m in meta if m[3])


def regs(tok):
    """Register numbers named by an operand token such as v12, v[50:51] (VGPRs only)."""
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]", tok):
        out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\bv(\d+)\b", tok):
        out.add(int(m.group(1)))
    return out


def _addr_index(lines):
    addr_of = {}
    for idx, ln in enumerate(lines):
        m = re.search(r"//\s*([0-9A-Fa-f]{8,16}):", ln)
        if m:
            addr_of.setdefault(int(m.group(1), 16), idx)
    return addr_of


def _branch_target(lines, k, addr_of):
    """(kind, line index of the target or None) of the branch instruction on line k; kind 'branch' / 'cbranch...'."""
    t = lines[k].strip().split("//")[0].strip()
    mb = re.match(r"s_(c?branch\w*)\s+(\d+)", t)
    if not mb:
        return None, None
    am = re.search(r"//\s*([0-9A-Fa-f]{8,16}):", lines[k])
    off = int(mb.group(2))
    off = off - 65536 if off >= 32768 else off
    tgt = addr_of.get(int(am.group(1), 16) + 4 + 4 * off) if am else None
    return mb.group(1), tgt


def check(text):
    """Every group of hand-issued ds_read_b64 (4, 9 or 16 in a row, kernels.hpp Taps<N>): on EVERY control-flow path from
    the group to the next `s_waitcnt lgkmcnt(0)` (the software-pipelined loop waits for a sample's reads after its
    back edge) no instruction may name one of the group's destination registers."""
    n_groups, bad = 0, []
    by_size = {}
    kernel = "?"
    lines = text.splitlines()
    addr_of = _addr_index(lines)
    i = 0
    while i < len(lines):
        ln = lines[i]
        m = re.match(r"^[0-9a-f]+ <(.+)>:", ln)
        if m:
            kernel = m.group(1)
        ins = ln.strip().split("//")[0].strip()
        if ins.startswith("ds_read_b64") and "k_sweep" in kernel:
            dst = set()
            j = i
            while j < len(lines) and lines[j].strip().startswith("ds_read_b64"):
                ops = lines[j].strip().split("//")[0].split(None, 1)[1]
                dst |= regs(ops.split(",")[0])
                j += 1
            if j - i in (4, 9, 16):  # Taps<2> / Taps<3> / Taps<4>
                n_groups += 1
                by_size[j - i] = by_size.get(j - i, 0) + 1
                work, seen = [j], set()
                while work:
                    k = work.pop()
                    steps = 0
                    while k < len(lines) and k not in seen and steps < 6000:
                        seen.add(k)
                        steps += 1
                        t = lines[k].strip().split("//")[0].strip()
                        if re.match(r"^[0-9a-f]+ <", lines[k]) or t.startswith("s_endpgm"):
                            bad.append((kernel, "end of kernel before the wait"))
                            break
                        if t.startswith("s_waitcnt") and "lgkmcnt(0)" in t:
                            break  # in-order return: this group's reads have landed
                        if t and not t.startswith(("s_nop", ";")):
                            ops = t.split(None, 1)[1] if " " in t else ""
                            if regs(ops) & dst:
                                bad.append((kernel, t))
                                break
                        kind, tgt = _branch_target(lines, k, addr_of)
                        if kind is not None:
                            if tgt is None:
                                bad.append((kernel, "branch to an unknown address before the wait: " + t))
                                break
                            work.append(tgt)
                            if kind == "branch":
                                break  # unconditional: no fall-through
                        k += 1
            i = j
            continue
        i += 1
    check.by_size = by_size
    return n_groups, bad


def sregs(tok):
    """SGPR numbers named by an operand token such as s12, s[36:43]."""
    out = set()
    for m in re.finditer(r"\bs\[(\d+):(\d+)\]", tok):
        out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\bs(\d+)\b", tok):
        out.add(int(m.group(1)))
    return out


def check_scalar_loads(text):
    """The point data of k_sweep come through hand-issued `s_load_dwordx8` (kernels.hpp spt_load) whose completion the
    compiler does not track: nothing may name the destination SGPRs before an `s_waitcnt lgkmcnt(0)` is reached -- on
    the fall-through path and on the target of every forward branch taken before that wait."""
    lines = text.splitlines()
    addr_of = {}
    for idx, ln in enumerate(lines):
        m = re.search(r"//\s*([0-9A-Fa-f]{8,16}):", ln)
        if m:
            addr_of.setdefault(int(m.group(1), 16), idx)
    n_loads, bad = 0, []
    kernel = "?"
    for i, ln in enumerate(lines):
        m = re.match(r"^[0-9a-f]+ <(.+)>:", ln)
        if m:
            kernel = m.group(1)
        ins = ln.strip().split("//")[0].strip()
        if not (ins.startswith("s_load_dwordx8") and "k_sweep" in kernel):
            continue
        dst = sregs(ins.split(None, 1)[1].split(",")[0])
        n_loads += 1
        work, seen = [i + 1], set()
        while work:
            k = work.pop()
            steps = 0
            while k < len(lines) and k not in seen and steps < 4000:
                seen.add(k)
                steps += 1
                t = lines[k].strip().split("//")[0].strip()
                if re.match(r"^[0-9a-f]+ <", lines[k]) or t.startswith("s_endpgm"):
                    break
                if t.startswith("s_waitcnt") and "lgkmcnt(0)" in t:
                    break
                if t and not t.startswith((";", "s_nop")) and " " in t:
                    if sregs(t.split(None, 1)[1]) & dst:
                        bad.append((kernel, t))
                        break
                mb = re.match(r"s_(c?branch\w*)\s+(\d+)", t)
                if mb:
                    am = re.search(r"//\s*([0-9A-Fa-f]{8,16}):", lines[k])
                    off = int(mb.group(2))
                    off = off - 65536 if off >= 32768 else off
                    if am:
                        tgt = addr_of.get(int(am.group(1), 16) + 4 + 4 * off)
                        if tgt is not None and off > 0:
                            work.append(tgt)
                        elif off <= 0 and mb.group(1) == "branch":
                            break  # loop back edge: the wait at the end of the point was on the way
                    if mb.group(1) == "branch":
                        break
                k += 1
    return n_loads, bad


def main(lib):
    n_k, lds, vgprs, spilled = check_static_lds(lib)
    print(f"[check_isa] ok: {n_k} k_sweep kernels, static LDS <= {lds} B (+ {MAX_DYNAMIC_LDS} dynamic <= "
          f"{LDS_PER_WORKGROUP}), <= {vgprs} VGPRs, {spilled} with VGPR spills or scratch")
    text = disassemble(lib)
    n_s, bad_s = check_scalar_loads(text)
    if bad_s:
        for k, t in bad_s[:20]:
            print("check_isa: SGPR of a pending scalar point load named before the wait:", k, "|", t, file=sys.stderr)
        raise SystemExit(f"check_isa: {len(bad_s)} scalar-load violation(s)")
    print(f"[check_isa] ok: {n_s} s_load_dwordx8 in k_sweep, none has its SGPRs named before an lgkmcnt(0) wait")
    n, bad = check(text)
    if n == 0:
        raise SystemExit("check_isa: no hand-issued ds_read_b64 group found in k_sweep (disassembly format changed?)")
    if bad:
        for k, t in bad[:20]:
            print("check_isa: tap register touched between the reads and their wait:", k, "|", t, file=sys.stderr)
        raise SystemExit(f"check_isa: {len(bad)} violation(s)")
    sizes = ", ".join(f"{v} of {k} reads" for k, v in sorted(getattr(check, "by_size", {}).items()))
    if not getattr(check, "by_size", {}).get(16):
        raise SystemExit("check_isa: no 16-read group (the cubic gather, Taps<4>) found in k_sweep")
    print(f"[check_isa] ok: {n} hand-issued LDS read groups ({sizes}), none has its registers touched before the wait")


if __name__ == "__main__":
    main(sys.argv[1])
