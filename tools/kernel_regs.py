"""Print VGPR/SGPR/scratch/spill counts per kernel from the engine's assembly (make -C csrc asm)."""
import re, sys, os
d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "climateparameterizations.jl_amd", "csrc", "_build")
s = "".join(open(os.path.join(d, f)).read() for f in ("engine_tile16.s", "engine_regtile.s") if os.path.exists(os.path.join(d, f)))
for b in s.split('  - .agpr_count:')[1:]:
    name = re.search(r'\.name:\s+(\S+)', b).group(1)
    g = lambda k: re.search(r'\.%s:\s+(\S+)' % k, b).group(1)
    if len(sys.argv) > 1 and not any(a in name for a in sys.argv[1:]):
        continue
    print(name[:52], 'vgpr', g('vgpr_count'), 'sgpr', g('sgpr_count'), 'scratch', g('private_segment_fixed_size'),
          'vspill', g('vgpr_spill_count'), 'sspill', g('sgpr_spill_count'))
