#!/usr/bin/env python3
"""One-off refactor helper: turn __global__ kernels of a file into __device__ bodies launched through
ZK_LAUNCHB (launch.hpp).  usage: convert_kernels.py file.hip [kernel ...]  (no kernels = all in file)"""
import re, sys
path = sys.argv[1]; only = set(sys.argv[2:])
s = open(path).read()
bounds = {}
def repl(m):
    n, name = m.group(1), m.group(2)
    if only and name not in only: return m.group(0)
    bounds[name] = n
    rest = m.group(3)
    return f"__device__ __forceinline__ void {name}(uint32_t zk_nblk" + (")" if rest.strip().startswith(")") else ", ")
s2 = re.sub(r"__global__ void __launch_bounds__\((\w+)\) (k_\w+)\((\s*\)?)", lambda m: repl(m) if True else "", s)
# the regex consumed an optional ')' for zero-arg kernels only
s = s2
if not only:
    s = s.replace("gridDim.x", "zk_nblk")
# call sites
out = []; i = 0
while True:
    j = s.find("ZK_LAUNCH(", i)
    if j < 0: out.append(s[i:]); break
    # skip the macro definition / comments
    out.append(s[i:j])
    k = j + len("ZK_LAUNCH("); depth = 1; args = []; cur = ""
    while depth:
        c = s[k]
        if c in "([{": depth += 1
        elif c in ")]}": depth -= 1
        if depth == 0: break
        if c == "," and depth == 1: args.append(cur); cur = ""
        else: cur += c
        k += 1
    args.append(cur)
    kern = args[3].strip() if len(args) > 5 else ""
    base = re.sub(r"[(<].*", "", kern.strip("() "))
    base = re.match(r"\(?\s*(k_\w+)", kern).group(1) if re.match(r"\(?\s*(k_\w+)", kern) else ""
    if base in bounds:
        blk = args[5].strip()
        m = re.fullmatch(r"dim3\((\d+)\)", blk)
        n = m.group(1) if m else bounds[base]
        if not m: print(f"NOTE {path}: block arg '{blk}' for {base}: using launch bound {n}", file=sys.stderr)
        new = "ZK_LAUNCHB(" + ",".join([args[0], args[1], args[2], " " + n, args[3], args[4], args[6]] + args[7:]) + ")"
        out.append(new)
    else:
        out.append(s[j:k + 1])
    i = k + 1
open(path, "w").write("".join(out))
print(path, "converted:", ", ".join(f"{k}[{v}]" for k, v in bounds.items()))
