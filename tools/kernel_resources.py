"""CPU: per-kernel register / LDS / spill figures of a built library, read from the gfx950 code object's metadata.
Usage: python tools/kernel_resources.py [library.so] [name regex]"""
import os, re, subprocess, sys, tempfile

so = sys.argv[1] if len(sys.argv) > 1 else "libcimbar_amd/libcimbar_hip.so"
pat = re.compile(sys.argv[2] if len(sys.argv) > 2 else ".")
B = "/opt/rocm/lib/llvm/bin/"
with tempfile.TemporaryDirectory() as t:
    subprocess.run([B + "llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", so, t + "/fat.bin"], check=True)
    subprocess.run([B + "clang-offload-bundler", "--type=o", "--input=" + t + "/fat.bin", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                    "--output=" + t + "/co.elf", "--unbundle"], check=True)
    txt = subprocess.run([B + "llvm-readelf", "--notes", t + "/co.elf"], capture_output=True, text=True, check=True).stdout
rows = []
for blk in txt.split("  - .agpr_count:")[1:]:
    g = lambda k: (re.search(r"\." + k + r":\s*(\S+)", blk) or [None, "?"])[1]
    rows.append((g("name"), g("vgpr_count"), g("sgpr_count"), g("group_segment_fixed_size"), g("private_segment_fixed_size"), g("vgpr_spill_count"), g("sgpr_spill_count")))
names = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.split("\n")
for r, nm in zip(rows, names):
    nm = nm.replace("(anonymous namespace)::", "").split("(")[0]
    if pat.search(nm):
        print(f"{nm:58s} vgpr {r[1]:>4} sgpr {r[2]:>4} lds {r[3]:>6} scratch {r[4]:>5} vspill {r[5]:>3} sspill {r[6]:>3}")
