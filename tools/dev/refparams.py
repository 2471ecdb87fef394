#!/usr/bin/env python3
"""Second step of tools/dev/batchify.py: struct-typed parameters of the `*_body` functions become const references.  A by-value copy
of an argument struct that is indexed with a run-time value (MergeArgs::in[cls] ...) lands in scratch memory (504 B/lane measured);
a reference into the kernarg segment does not."""
import re, sys
SCALAR = {"int", "float", "double", "unsigned", "long", "bool", "char", "short", "size_t", "uint32_t", "uint64_t", "unsigned long long", "long long", "unsigned int"}
HEAD = re.compile(r"__device__ __forceinline__ void (k_\w+_body)\s*\(")

def split_params(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "(<[":
            depth += 1
        elif ch in ")>]":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur); cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return out

def fix(path):
    s = open(path).read()
    pos, out, n = 0, [], 0
    while True:
        m = HEAD.search(s, pos)
        if not m:
            break
        i = m.end() - 1
        depth = 0
        j = i
        while True:
            if s[j] == "(": depth += 1
            elif s[j] == ")":
                depth -= 1
                if depth == 0: break
            j += 1
        params = split_params(s[i + 1:j])
        new = []
        for p in params:
            t = p.strip()
            mm = re.match(r"^(.*?)(\w+)$", t, re.S)
            if not mm or not mm.group(1).strip():
                new.append(p)
                continue
            ty = mm.group(1).strip()
            base = ty.replace("const", "").strip()
            if "*" in ty or "&" in ty or base in SCALAR or ty.startswith("const "):
                new.append(p)
            else:
                lead = p[:len(p) - len(p.lstrip())]
                new.append(f"{lead}const {ty}& {mm.group(2)}")
                n += 1
        out.append(s[pos:i + 1] + ",".join(new))
        pos = j
    out.append(s[pos:])
    open(path, "w").write("".join(out))
    return n

for f in sys.argv[1:]:
    print(f, fix(f))
