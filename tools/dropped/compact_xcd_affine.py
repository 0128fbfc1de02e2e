import sys, os
d = sys.argv[1]
p = os.path.join(d, 'compaction.hpp')
s = open(p).read()
old = '''    const int t = blockIdx.x - h.blocks;
'''
new = '''    int t = blockIdx.x - h.blocks;
    if (a.xcd_affine) { t += (h.blocks * (b + 1)) & 7; if (t >= a.T) t -= a.T; }   // the XCD of the scan block that wrote this tile (b*T + t) % 8
'''
assert old in s
s = s.replace(old, new)
s = s.replace('''    int fuse_sub;            // 1:''', '''    int xcd_affine;
    int fuse_sub;            // 1:''', 1)
open(p, 'w').write(s)
p = os.path.join(d, 'pvnet_vote.hip')
s = open(p).read()
old = '''    m.want_draws = (f.can_subsample && !m.fuse_sub) ? 1 : 0;'''
assert old in s
s = s.replace(old, '''    m.xcd_affine = tuning_int("PVV_XCD_AFFINE", 0);
''' + old)
open(p, 'w').write(s)
