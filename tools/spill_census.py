"""CPU: where the SGPR spills of one kernel sit. The compiler spills scalar registers into lanes of a few dedicated VGPRs (v_writelane_b32 vN, sX, <lane
constant>; v_readlane_b32 sX, vN, <lane constant>); the spill VGPRs are the destinations of v_writelane with a constant lane. The script finds the
kernel's loops (backward branches), and for each prints its length, its scalar instructions and how many spill stores / reloads it holds -- lane reads
with a register selector or from other VGPRs are the algorithm's own and are listed apart.
Usage: python tools/spill_census.py [library.so] <demangled-name regex>     e.g.  'k_png_inflate<8192, true>'"""
import collections, re, subprocess, sys, tempfile

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
        cur = m.group(1).replace("(anonymous namespace)::", "").split("(")[0]
        continue
    m = re.match(r"^\s+(\S+)\s+(.*?)\s*//\s*([0-9A-F]+):", line)
    if cur and m:
        body[cur].append((int(m.group(3), 16), m.group(1), m.group(2), line))

for name, ins in body.items():
    if not pat.search(name):
        continue
    spill_v = {a.split(",")[0].strip() for _, op, a, _ in ins if op.startswith("v_writelane") and re.search(r",\s*\d+$", a)}
    is_store = lambda op, a: op.startswith("v_writelane") and a.split(",")[0].strip() in spill_v
    is_reload = lambda op, a: bool(op.startswith("v_readlane") and a.split(",")[1].strip() in spill_v and re.search(r",\s*\d+$", a))
    is_own = lambda op, a: (op.startswith("v_readlane") or op.startswith("v_writelane")) and not is_store(op, a) and not is_reload(op, a)
    idx = {a: k for k, (a, _, _, _) in enumerate(ins)}
    base = ins[0][0]
    loops = set()
    for k, (a, op, args, line) in enumerate(ins):
        if op.startswith("s_cbranch") or op == "s_branch":
            m = re.search(r"<.*\+0x([0-9a-f]+)>", line)
            if m and base + int(m.group(1), 16) <= a and base + int(m.group(1), 16) in idx:
                loops.add((idx[base + int(m.group(1), 16)], k))
    print(f"{name}: {len(ins)} instructions, spill VGPRs {sorted(spill_v)}, {sum(is_store(o, a) for _, o, a, _ in ins)} spill stores, "
          f"{sum(is_reload(o, a) for _, o, a, _ in ins)} reloads, {sum(is_own(o, a) for _, o, a, _ in ins)} lane reads / writes of the algorithm's own")
    # innermost first; a loop is "innermost" when no other loop lies strictly inside it
    inner = [l for l in loops if not any(o != l and o[0] >= l[0] and o[1] <= l[1] for o in loops)]
    print(f"  {len(loops)} loops, {len(inner)} innermost. Innermost loops (no loop inside them), by length:")
    for s, e in sorted(inner, key=lambda l: l[0] - l[1]):
        seg = ins[s:e + 1]
        print(f"    [{s:5d} .. {e:5d}] {e - s + 1:5d} instructions, {sum(o.startswith('s_') for _, o, _, _ in seg):4d} scalar, spill stores {sum(is_store(o, a) for _, o, a, _ in seg):3d}, "
              f"reloads {sum(is_reload(o, a) for _, o, a, _ in seg):3d}, own lane traffic {sum(is_own(o, a) for _, o, a, _ in seg):3d}")
    print("  Enclosing loops, by length:")
    for s, e in sorted(loops - set(inner), key=lambda l: l[1] - l[0]):
        seg = ins[s:e + 1]
        print(f"    [{s:5d} .. {e:5d}] {e - s + 1:5d} instructions, spill stores {sum(is_store(o, a) for _, o, a, _ in seg):3d}, reloads {sum(is_reload(o, a) for _, o, a, _ in seg):3d}")
