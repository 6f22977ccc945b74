"""VGPRs / spilled registers / scratch bytes of every kernel instance in the shipped gfx950 code object (no GPU needed).
    python tools/kernel_regs.py [substring ...]        # e.g. python tools/kernel_regs.py k_adapt k_aem
Writes nothing; profiles/rNN_scratch_counts.txt is this table for all instances."""
import os, re, shutil, subprocess, sys, tempfile
LLVM_BIN = "/opt/rocm/lib/llvm/bin"
LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tinyda_amd", "lib", "libtinyda_hip.so")

def metadata(lib=LIB):
    tmp = tempfile.mkdtemp(prefix="tda_co_")
    try:
        so = os.path.join(tmp, "lib.so")
        shutil.copy(lib, so)
        subprocess.run([os.path.join(LLVM_BIN, "llvm-objdump"), "--offloading", so], cwd=tmp, check=True, capture_output=True)
        co = [f for f in os.listdir(tmp) if "gfx950" in f]
        notes = subprocess.run([os.path.join(LLVM_BIN, "llvm-readelf"), "--notes", os.path.join(tmp, co[0])], check=True, capture_output=True, text=True).stdout
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    meta = {}
    for m in re.finditer(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)", notes):
        meta[m.group(1)] = dict(scratch_bytes=int(m.group(2)), vgprs=int(m.group(3)), spilled=int(m.group(4)))
    return meta

if __name__ == "__main__":
    pats = sys.argv[1:]
    filt = shutil.which("c++filt")
    dem = lambda n: subprocess.run([filt, n], capture_output=True, text=True).stdout.strip() if filt else n
    for name, md in sorted(metadata().items()):
        if pats and not any(p in name for p in pats):
            continue
        print("%-90s vgprs %3d  spilled %3d  scratch %5d B" % (dem(name)[:90], md["vgprs"], md["spilled"], md["scratch_bytes"]))
