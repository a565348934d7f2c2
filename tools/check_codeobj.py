#!/usr/bin/env python3
"""Guards against a ROCm 7.2 / LLVM 22 code-generation hazard on gfx950: inside a non-kernel device
function, a branch longer than the SOPP range is relaxed through `s_getpc_b64 s[30:31]`, destroying the
return address (the wave then never returns).  Usage: check_codeobj.py <file.s> ; exit 1 if any
out-of-line function contains a relaxed branch or is larger than a safety bound."""
import re
import sys

LIMIT_INSTR = 14000  # ~100 KiB of the 128 KiB branch range


def main(path):
    s = open(path).read()
    kernels = set(re.findall(r"\.amdhsa_kernel (\S+)", s))
    bad = 0
    for m in re.finditer(r"^(_Z\w+):.*?\n(.*?)^\.Lfunc_end\d+:", s, re.S | re.M):
        name, body = m.group(1), m.group(2)
        if name in kernels:
            continue
        n = sum(1 for l in body.split("\n") if l.startswith("\t") and not l.startswith("\t.") and not l.strip().startswith(";"))
        relaxed = ".Lpost_getpc" in body
        if relaxed or n > LIMIT_INSTR:
            print("UNSAFE out-of-line function (%d instr, relaxed_branch=%s): %s" % (n, relaxed, name))
            bad += 1
    print("check_codeobj: %d kernels, %s" % (len(kernels), "FAILED" if bad else "ok"))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1]))
