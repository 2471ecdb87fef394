#!/usr/bin/env python3
"""One-off source transformation (round 3): every non-template __global__ kernel of the hot-path files becomes a __device__ body +
SCAL_KERNEL (csrc/batch.hpp), its launches become SCAL_LAUNCH, and the stream operations go through the recordable wrappers.
Kept for the record; the transformed sources are what is committed."""
import re
import sys

FILES = ["features.hip", "mapping.hip", "odometry.hip", "voxel.hip", "radix_sort.hip", "scancontext.hip", "lm_dev.hpp"]
OPS = [("hipEventRecord(", "op_event_record("), ("hipStreamWaitEvent(", "op_stream_wait_event("), ("hipMemcpyAsync(", "op_memcpy_async("),
       ("hipMemsetAsync(", "op_memset_async("), ("hipEventSynchronize(", "op_event_synchronize("), ("hipStreamSynchronize(", "op_stream_synchronize("),
       ("hipEventQuery(", "op_event_query(")]
HEAD = re.compile(r"(?:static\s+)?__global__\s+void\s+(?:__launch_bounds__\(([^)]*)\)\s+)?(k_\w+)\s*\(")


def match_close(s, i, open_ch, close_ch):
    depth = 0
    while i < len(s):
        c = s[i]
        if c == open_ch:
            depth += 1
        elif c == close_ch:
            depth -= 1
            if depth == 0:
                return i
        elif c == '"':
            i = s.index('"', i + 1)
        elif c == "'" and s[i + 2] == "'":
            i += 2
        elif s.startswith("//", i):
            i = s.index("\n", i)
        elif s.startswith("/*", i):
            i = s.index("*/", i) + 1
        i += 1
    raise ValueError("unbalanced")


def transform(path):
    s = open(path).read()
    names = []
    pos = 0
    out = []
    while True:
        m = HEAD.search(s, pos)
        if not m:
            break
        # skip template kernels (handled by hand)
        line_start = s.rfind("\n", 0, m.start()) + 1
        prev = s[max(0, line_start - 200):line_start]
        prev_line = prev.rstrip().splitlines()[-1] if prev.strip() else ""
        if prev_line.strip().startswith("template"):
            out.append(s[pos:m.end()])
            pos = m.end()
            continue
        bounds, name = m.group(1), m.group(2)
        par_open = m.end() - 1
        par_close = match_close(s, par_open, "(", ")")
        brace_open = s.index("{", par_close)
        if s[par_close + 1:brace_open].strip():
            raise ValueError(f"{path}: unexpected text between ) and {{ of {name}")
        brace_close = match_close(s, brace_open, "{", "}")
        out.append(s[pos:m.start()])
        out.append(f"__device__ __forceinline__ void {name}_body" + s[par_open:brace_close + 1])
        out.append(f"\nSCAL_KERNEL({bounds or 1024}, {name})")
        pos = brace_close + 1
        names.append(name)
    out.append(s[pos:])
    s = "".join(out)
    return s, names


def main(root):
    all_names = []
    texts = {}
    for f in FILES:
        s, names = transform(f"{root}/{f}")
        texts[f] = s
        all_names += names
    for f in FILES + ["common.hpp"]:
        s = texts.get(f) or open(f"{root}/{f}").read()
        # launches of transformed kernels
        def fix(m):
            return "SCAL_LAUNCH(" if m.group(2) in all_names else m.group(0)
        s = re.sub(r"SCAL_LAUNCH_PROF\((\s*[^,]+,\s*)(k_\w+)(?=,)", lambda m: ("SCAL_LAUNCH(" + m.group(1) + m.group(2)) if m.group(2) in all_names else m.group(0), s)
        for a, b in OPS:
            s = s.replace(a, b)
        open(f"{root}/{f}", "w").write(s)
    print(len(all_names), "kernels:", " ".join(all_names))


if __name__ == "__main__":
    main(sys.argv[1])
