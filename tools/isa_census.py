"""CPU: static instruction census of one kernel of a built library (gfx950 code object disassembled with llvm-objdump): how many scalar / vector / LDS /
memory instructions, the most frequent opcodes, and how the scalar ones split into exec-mask bookkeeping, branches, compares and arithmetic.
Usage: python tools/isa_census.py [library.so] <demangled-name regex>     e.g.  'm68.*k_flood3<4095, true>'"""
import collections, os, re, subprocess, sys, tempfile

so = sys.argv[1] if len(sys.argv) > 2 else "libcimbar_amd/libcimbar_hip.so"
pat = re.compile(sys.argv[-1])
B = "/opt/rocm/lib/llvm/bin/"
with tempfile.TemporaryDirectory() as t:
    subprocess.run([B + "llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", so, t + "/fat.bin"], check=True)
    subprocess.run([B + "clang-offload-bundler", "--type=o", "--input=" + t + "/fat.bin", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                    "--output=" + t + "/co.elf", "--unbundle"], check=True)
    txt = subprocess.run([B + "llvm-objdump", "-d", "--demangle", t + "/co.elf"], capture_output=True, text=True).stdout
cur, body = None, collections.defaultdict(list)
for line in txt.splitlines():
    m = re.match(r"^[0-9a-f]+ <(.*)>:$", line)
    if m:
        cur = m.group(1).replace("(anonymous namespace)::", "")
        continue
    m = re.match(r"^\s+([a-z_0-9]+)\s", line)
    if cur and m:
        body[cur].append(m.group(1))
for name, ops in body.items():
    short = name.split("(")[0]
    if not pat.search(short):
        continue
    c = collections.Counter(ops)
    grp = collections.Counter()
    for op, k in c.items():
        if op.startswith("s_"):
            if "exec" in op or op in ("s_and_b64", "s_or_b64", "s_andn2_b64", "s_mov_b64", "s_xor_b64", "s_cselect_b64", "s_not_b64"):
                grp["scalar: 64-bit lane-mask / exec bookkeeping"] += k
            elif op.startswith("s_cbranch") or op == "s_branch":
                grp["scalar: branches"] += k
            elif op.startswith("s_cmp") or op.startswith("s_bitcmp"):
                grp["scalar: compares"] += k
            elif op in ("s_waitcnt", "s_nop", "s_barrier", "s_setprio", "s_sleep"):
                grp["scalar: waitcnt / nop / barrier"] += k
            elif op.startswith("s_load") or op.startswith("s_buffer"):
                grp["scalar: memory"] += k
            else:
                grp["scalar: arithmetic / moves"] += k
        elif op.startswith("v_readlane") or op.startswith("v_readfirstlane") or op.startswith("v_writelane"):
            grp["vector <-> scalar lane traffic"] += k
        elif op.startswith("v_cmp"):
            grp["vector: compares (lane masks)"] += k
        elif op.startswith("v_"):
            grp["vector: other"] += k
        elif op.startswith("ds_"):
            grp["LDS"] += k
        elif op.startswith("global_") or op.startswith("flat_") or op.startswith("buffer_") or op.startswith("scratch_"):
            grp["global / scratch memory"] += k
        else:
            grp["other"] += k
    print(f"{short}: {len(ops)} instructions")
    for g, k in sorted(grp.items(), key=lambda x: -x[1]):
        print(f"    {k:5d}  {g}")
    print("    top opcodes: " + ", ".join(f"{op} {k}" for op, k in c.most_common(24)))
