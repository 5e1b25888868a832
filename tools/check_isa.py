"""Compile every kernel source to gfx950 ISA and fail on what has bitten this library:
  * scratch memory (private segment > 0): a run-time index into a register array — spills are never intended here;
  * relative register addressing (s_set_gpr_idx / v_movrel): the same pattern compiled another way; an index beyond the
    array (even behind an `if`) died with a memory-aperture violation on MI355X (mix_feature_nhwc, C = 6000).
Usage: python tools/check_isa.py        (CPU only; ~2 minutes)"""
import glob, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
bad = 0
for src in sorted(glob.glob(os.path.join(ROOT, "cv_a-fan_amd", "csrc", "*.hip"))):
    with tempfile.NamedTemporaryFile(suffix=".s") as f:
        subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math",
                        "-w", "-S", "--cuda-device-only", "-o", f.name, src], check=True)
        isa = open(f.name).read()
    scratch = re.findall(r"\.set (\S+)\.private_seg_size, ([1-9]\d*)", isa)
    rel = len(re.findall(r"s_set_gpr_idx_on|v_movrel", isa))
    print(f"{os.path.basename(src):28s} scratch kernels: {len(scratch)}  relative-addressing instructions: {rel}")
    for name, size in scratch:
        print(f"    {name}: {size} bytes of scratch")
    bad += len(scratch) + rel
sys.exit(1 if bad else 0)
