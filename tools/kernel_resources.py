"""Register / scratch / LDS budget of every kernel in a gfx950 assembly file (hipcc --save-temps):
    python tools/kernel_resources.py file.s [substring]
Used by tests/test_build_guards.py: the hand-written node loop of wf_extend is correct under any register allocation
(clobber list), but it is only FAST while the kernel around it stays at 64 VGPRs without scratch."""
import re, subprocess, sys

def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
        return out[:len(names)]
    except OSError:
        return names

def kernels(path):
    s = open(path).read()
    meta = s[s.index("amdhsa.kernels:"):]
    blocks = re.split(r"\n  - \.agpr_count:", meta)[1:]
    rows = []
    for b in blocks:
        g = lambda k: re.search(r"\." + k + r":\s+(\S+)", b).group(1)
        rows.append(dict(name=g("name"), vgpr=int(g("vgpr_count")), sgpr=int(g("sgpr_count")), scratch=int(g("private_segment_fixed_size")),
                         lds=int(g("group_segment_fixed_size")), agpr=int(re.match(r"\s*(\d+)", b).group(1))))
    for r, d in zip(rows, demangle([r["name"] for r in rows])):
        r["demangled"] = d
    return rows

if __name__ == "__main__":
    for r in kernels(sys.argv[1]):
        if len(sys.argv) > 2 and sys.argv[2] not in r["demangled"]: continue
        print(f"{r['vgpr']:4d} vgpr {r['agpr']:3d} agpr {r['sgpr']:4d} sgpr {r['scratch']:5d} scratch {r['lds']:6d} lds  {r['demangled'][:120]}")
