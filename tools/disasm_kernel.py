"""CPU: disassembly of one kernel of a built library (gfx950 code object, llvm-objdump) on stdout.
Usage: python tools/disasm_kernel.py [library.so] <demangled-name regex>"""
import os, re, subprocess, sys, tempfile
so = sys.argv[1] if len(sys.argv) > 2 else "libcimbar_amd/libcimbar_hip.so"
pat = re.compile(sys.argv[-1])
B = "/opt/rocm/lib/llvm/bin/"
with tempfile.TemporaryDirectory() as t:
    subprocess.run([B + "llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", so, t + "/fat.bin"], check=True)
    subprocess.run([B + "clang-offload-bundler", "--type=o", "--input=" + t + "/fat.bin", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                    "--output=" + t + "/co.elf", "--unbundle"], check=True)
    txt = subprocess.run([B + "llvm-objdump", "-d", "--demangle", "--no-show-raw-insn", t + "/co.elf"], capture_output=True, text=True).stdout
on = False
for line in txt.splitlines():
    m = re.match(r"^[0-9a-f]+ <(.*)>:$", line)
    if m:
        on = bool(pat.search(m.group(1).replace("(anonymous namespace)::", "").split("(")[0]))
    if on:
        print(line)
