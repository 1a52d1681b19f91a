"""Diagnostic: instruction mix and a compact op trace of the innermost (stage) loop of a kernel in the generated assembly.
usage: loop_mix.py [asm file] [mangled-name prefix]   (make -C climateparameterizations.jl_amd/csrc asm first)"""
import collections, re, sys, textwrap, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = [a for a in sys.argv[1:] if not a.startswith("-")]
path = args[0] if len(args) > 0 else os.path.join(ROOT, "climateparameterizations.jl_amd/csrc/_build/engine_regtile.s")
name = args[1] if len(args) > 1 else "_Z17rt_adjoint_kernelILi2ELb1ELb1EEv"
s = open(path).read()
start = s.index(name); start = s.index(":", start)
end = s.index(".Lfunc_end", start)
k = s[start:end].split("\n")
depth = [None] * len(k)
cur = 0
# innermost loop = lines whose block comment says "Depth=<max>"
maxd = max(int(m.group(1)) for l in k for m in [re.search(r"Depth=(\d+)", l)] if m)
inner = [i for i, l in enumerate(k) if re.search(r"Depth=%d" % maxd, l)]
lo, hi = inner[0], inner[-1]
# extend hi to the end of the last block of the loop (next label)
j = hi + 1
while j < len(k) and not re.match(r"\.LBB\d+_\d+:", k[j]): j += 1
body = [l.strip() for l in k[lo:j]]
body = [l for l in body if l and not l.startswith((".", ";"))]


def short(l):
    op = l.split()[0]
    if "mfma" in op: return "M"
    if op.startswith(("v_exp", "v_log")): return "E"
    if op.startswith("v_rcp"): return "R"
    if op.startswith("ds_read"): return "d"
    if op.startswith("ds_write"): return "w"
    if op.startswith("s_waitcnt"): return "V" if "vmcnt" in l else "|"
    if op.startswith("s_nop"): return "n"
    if op.startswith("scratch_load"): return "L"
    if op.startswith("scratch_store"): return "S"
    if op.startswith("global_load"): return "G"
    if op.startswith("global_store"): return "T"
    if op.startswith("v_accvgpr"): return "a"
    if op.startswith("v_"): return "."
    return "s"


tr = "".join(short(l) for l in body)
print("%s: innermost loop (depth %d), %d instructions" % (name, maxd, len(body)))
print(dict(collections.Counter(tr).most_common()))
print("legend: M mfma  . valu  E exp/log  R rcp  a accvgpr move  d/w LDS read/write  | lgkm wait  V vm wait  G/T global ld/st  L/S scratch ld/st  n nop  s scalar")
if "-q" not in sys.argv:
    print("\n".join(textwrap.wrap(tr, 200)))
ops = collections.Counter(l.split()[0] for l in body)
print(ops.most_common(25))
