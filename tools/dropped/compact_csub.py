"""Patch for tools/build_variant.sh --py: PVV_CSUB compaction blocks per tile in k_compact_hyp (block `part` of a tile takes the
rows part*256 + [0,256), + csub*256, ...; every part redoes the prefix).  Measured in round 4: see there.
usage: tools/build_variant.sh csub -DPVV_TUNING --py tools/dropped/compact_csub.py ; PVV_CSUB=1|2|4|8"""
import sys, os
d = sys.argv[1]
p = os.path.join(d, 'compaction.hpp')
s = open(p).read()
def rep(old, new):
    global s
    assert old in s, old
    s = s.replace(old, new)
rep('''    int fuse_sub;            // 1:''', '''    int csub;
    int fuse_sub;            // 1:''')
rep('''    const int t = blockIdx.x - h.blocks;
''', '''    const int t = (int)(blockIdx.x - h.blocks) / a.csub, part = (int)(blockIdx.x - h.blocks) - t * a.csub;
''')
rep('''    if (t != 0 && !(a.fuse_sub && t == a.T - 1) && nz == 0) return;''',
    '''    const bool reporter = part == 0 && (t == 0 || (a.fuse_sub && t == a.T - 1));
    if (!reporter && nz <= part * kBlock) return;''')
s = s.replace('if (t == 0 && threadIdx.x == 0)', 'if (t == 0 && part == 0 && threadIdx.x == 0)')
rep('if (t == a.T - 1 && threadIdx.x == 0) {', 'if (t == a.T - 1 && part == 0 && threadIdx.x == 0) {')
rep('for (int li = threadIdx.x; li < n; li += kBlock) {', 'for (int li = part * kBlock + threadIdx.x; li < n; li += a.csub * kBlock) {')
open(p, 'w').write(s)
p = os.path.join(d, 'pvnet_vote.hip')
s = open(p).read()
rep('''    m.want_draws = (f.can_subsample && !m.fuse_sub) ? 1 : 0;''', '''    m.csub = tuning_int("PVV_CSUB", 1);
    m.want_draws = (f.can_subsample && !m.fuse_sub) ? 1 : 0;''')
rep('dim3(L.T + f.h.blocks, p->B), dim3(kBlock), sizeof(int) * (size_t)L.T', 'dim3(L.T * f.m.csub + f.h.blocks, p->B), dim3(kBlock), sizeof(int) * (size_t)L.T')
open(p, 'w').write(s)
