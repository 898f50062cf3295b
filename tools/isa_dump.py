#!/usr/bin/env python3
"""tools/isa_dump.py LIB OUTDIR — one text file per kernel / device function of every gfx950 code object in LIB (llvm-objdump -d, addresses and
branch-target offsets stripped), so that two builds can be compared function by function with `diff -r`: the way a results-neutral source
change is shown to be ISA-neutral (or confined to the instructions it should touch) when there is no GPU at hand to run it.  Build-time
analysis only; nothing is executed."""
import hashlib, os, re, subprocess, sys, tempfile
LLVM = "/opt/rocm/lib/llvm/bin/"


def main(so, outdir):
    os.makedirs(outdir, exist_ok=True)
    with tempfile.TemporaryDirectory() as td:
        fat = os.path.join(td, "fat.bin")
        subprocess.run(["objcopy", "-O", "binary", "-j", ".hip_fatbin", so, fat], check=True)
        data = open(fat, "rb").read()
        offs = [m.start() for m in re.finditer(b"__CLANG_OFFLOAD_BUNDLE__", data)] + [len(data)]
        n = 0
        for i in range(len(offs) - 1):
            b, co = os.path.join(td, "b.bin"), os.path.join(td, "b.co")
            open(b, "wb").write(data[offs[i]:offs[i + 1]])
            if subprocess.run([LLVM + "clang-offload-bundler", "--unbundle", "--type=o", "--input=" + b, "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], capture_output=True).returncode:
                continue
            dis = subprocess.run([LLVM + "llvm-objdump", "-d", "--no-show-raw-insn", co], capture_output=True, text=True).stdout
            cur, body = None, []

            def flush():
                nonlocal n
                if cur is None:
                    return
                name = cur if len(cur) < 120 else cur[:80] + "_" + hashlib.sha1(cur.encode()).hexdigest()[:12]
                open(os.path.join(outdir, name + ".s"), "a").write("\n".join(body) + "\n")
                n += 1
            for l in dis.splitlines():
                m = re.match(r"^[0-9a-f]+ <([^>]+)>:$", l)
                if m:
                    flush()
                    cur, body = m.group(1), []
                    continue
                if cur is None:
                    continue
                l = re.sub(r"^\s*[0-9a-f]+:\s*", "", l)                 # instruction address
                l = re.sub(r"//.*$", "", l).rstrip()                      # trailing address comments
                l = re.sub(r"<[^>]+\+0x[0-9a-f]+>", "<target>", l)       # symbolic branch targets
                l = re.sub(r"(s_cbranch\w*|s_branch)\s+\S+", r"\1 <target>", l)
                if l:
                    body.append(l)
            flush()
    print("%d functions -> %s" % (n, outdir))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
