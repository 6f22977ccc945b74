"""Loop-level audit of the shipped gfx950 code object (no GPU needed): for every kernel, the loops of its disassembly (a backward
branch to a label = one loop) with what DESIGN.md §11.6 says to look for INSIDE a step loop -- `s_waitcnt vmcnt(0)` (on gfx9 the
counter is in-order over loads AND stores: a full wait drains every prefetch in flight), `scratch_` accesses (a spill reload is a
full wait too), barriers -- next to the matrix / vector / memory instruction counts of the body.

    python tools/loop_audit.py [substring ...]       e.g.  python tools/loop_audit.py k_mh_steps k_da_steps_r224

tests/test_code_object.py holds the step loops of the hot kernels to the counts this prints."""
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM_BIN = "/opt/rocm/lib/llvm/bin"
LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tinyda_amd", "lib", "libtinyda_hip.so")

_FUNC = re.compile(r"^([0-9a-f]{16}) <([^>]+)>:$")
_INSN = re.compile(r"^\t(\S+)\s*(.*?)\s*// ([0-9A-F]{12}):")
_BRANCH = re.compile(r"^s_c?branch\S*$")


def disassemble(lib=LIB):
    """{kernel symbol: [(address, mnemonic, operands)]} and {kernel symbol: {label: address}}"""
    tmp = tempfile.mkdtemp(prefix="tda_dis_")
    try:
        so = os.path.join(tmp, "lib.so")
        shutil.copy(lib, so)
        subprocess.run([os.path.join(LLVM_BIN, "llvm-objdump"), "--offloading", so], cwd=tmp, check=True, capture_output=True)
        co = [f for f in os.listdir(tmp) if "gfx950" in f]
        text = subprocess.run([os.path.join(LLVM_BIN, "llvm-objdump"), "-d", "--symbolize-operands", os.path.join(tmp, co[0])],
                              check=True, capture_output=True, text=True).stdout
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    kernels, labels, cur = {}, {}, None
    for ln in text.splitlines():
        m = _FUNC.match(ln)
        if m:
            name = m.group(2)
            if re.fullmatch(r"L\d+", name):
                if cur is not None:
                    labels[cur][name] = int(m.group(1), 16)
            else:
                cur = name
                kernels[cur], labels[cur] = [], {}
            continue
        m = _INSN.match(ln)
        if m and cur is not None:
            kernels[cur].append((int(m.group(3), 16), m.group(1), m.group(2)))
    return kernels, labels


def loops_of(insns, labs):
    """[(start address, end address (the backward branch), depth)] sorted outermost first"""
    out = []
    for addr, mn, ops in insns:
        if _BRANCH.match(mn):
            tgt = labs.get(ops.split()[0]) if ops else None
            if tgt is not None and tgt <= addr:
                out.append((tgt, addr))
    out = sorted(set(out), key=lambda l: (l[0], -l[1]))
    res = []
    for lo, hi in out:
        depth = sum(1 for a, b in out if a <= lo and hi <= b and (a, b) != (lo, hi))
        res.append((lo, hi, depth))
    return res


def _full_vm_wait(mn, ops):
    return mn == "s_waitcnt" and re.search(r"vmcnt\(0\)", ops) is not None


def body_stats(insns, lo, hi):
    body = [(a, mn, ops) for a, mn, ops in insns if lo <= a <= hi]
    return dict(
        instructions=len(body),
        mfma=sum(1 for _, mn, _ in body if mn.startswith("v_mfma")),
        valu=sum(1 for _, mn, _ in body if mn.startswith("v_") and not mn.startswith("v_mfma")),
        vmem_load=sum(1 for _, mn, _ in body if re.match(r"(global|buffer|flat)_load", mn)),
        vmem_store=sum(1 for _, mn, _ in body if re.match(r"(global|buffer|flat)_(store|atomic)", mn)),
        lds=sum(1 for _, mn, _ in body if mn.startswith("ds_")),
        barriers=sum(1 for _, mn, _ in body if mn == "s_barrier"),
        vmcnt0=sum(1 for _, mn, ops in body if _full_vm_wait(mn, ops)),
        waitcnt_vm=sum(1 for _, mn, ops in body if mn == "s_waitcnt" and "vmcnt" in ops),
        scratch=sum(1 for _, mn, _ in body if mn.startswith("scratch_")),
        flat=sum(1 for _, mn, _ in body if mn.startswith("flat_")),
    )


def step_loop(insns, labs):
    """The kernel's STEP loop: the outermost loop that contains matrix instructions (one trip = one Metropolis-Hastings step of the
    tile, or one coarse step of the hierarchy); kernels without matrix instructions: the outermost loop that stores to memory.
    Returns (lo, hi, stats) or None."""
    cands = []
    for lo, hi, depth in loops_of(insns, labs):
        st = body_stats(insns, lo, hi)
        cands.append((depth, -st["instructions"], lo, hi, st))
    for key in ("mfma", "vmem_store"):
        hit = sorted(c for c in cands if c[4][key] > 0)
        if hit:
            return hit[0][2], hit[0][3], hit[0][4]
    return None


def audit(patterns=None, lib=LIB):
    kernels, labels = disassemble(lib)
    res = {}
    for name, insns in kernels.items():
        if patterns and not any(p in name for p in patterns):
            continue
        whole = body_stats(insns, insns[0][0], insns[-1][0]) if insns else {}
        res[name] = dict(whole=whole, loops=[(lo, hi, d, body_stats(insns, lo, hi)) for lo, hi, d in loops_of(insns, labels[name])],
                         step=step_loop(insns, labels[name]))
    return res


if __name__ == "__main__":
    filt = shutil.which("c++filt")
    dem = lambda n: subprocess.run([filt, n], capture_output=True, text=True).stdout.strip() if filt else n
    for name, r in sorted(audit(sys.argv[1:]).items()):
        w = r["whole"]
        print("%s\n  kernel: %d instructions, %d mfma, %d vmcnt(0) of %d vm waits, %d scratch" % (dem(name)[:110], w["instructions"], w["mfma"], w["vmcnt0"], w["waitcnt_vm"], w["scratch"]))
        for lo, hi, d, st in r["loops"]:
            mark = " <- step loop" if r["step"] and (lo, hi) == r["step"][:2] else ""
            print("  %sloop %x..%x: %5d instr, mfma %4d, valu %5d, ld %3d, st %3d, lds %4d, barriers %2d, vmcnt(0) %2d / %2d vm waits, scratch %d, flat %d%s"
                  % ("  " * d, lo, hi, st["instructions"], st["mfma"], st["valu"], st["vmem_load"], st["vmem_store"], st["lds"], st["barriers"],
                     st["vmcnt0"], st["waitcnt_vm"], st["scratch"], st["flat"], mark))
