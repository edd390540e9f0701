"""Instruction counts of device kernels from a hipcc -S listing (development helper):
   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off --cuda-device-only -S -o /tmp/x.s columba_amd/csrc/columba_amd.hip
   python3 tools/isa_stats.py /tmp/x.s k_bfs_pass [--loops]"""
import re, sys
src = open(sys.argv[1]).read().split('\n')
pat = sys.argv[2]
starts = [i for i, l in enumerate(src) if re.match(r'^_Z\w+:', l) and pat in l.split(':')[0]]
for st in starts:
    en = st
    while 'codeLenInByte' not in src[en]:
        en += 1
    meta = {}
    for j in range(en, min(en + 40, len(src))):
        m = re.match(r'; (NumVgprs|ScratchSize|Occupancy|codeLenInByte|LDSByteSize)[: =]+(\d+)', src[j])
        if m:
            meta[m.group(1)] = int(m.group(2))
    body = src[st:en]
    ins = [l for l in body if l.startswith('\t') and not l.startswith('\t.') and not l.startswith('\t;')]
    print(src[st].split(':')[0][:100])
    print('  ', meta, '| instructions', len(ins), '| valu', sum(l.startswith('\tv_') for l in ins), '| 64-bit valu',
          sum(bool(re.match(r'\tv_\w+_[bui]64|\tv_lshl_add_u64', l)) for l in ins), '| salu', sum(l.startswith('\ts_') for l in ins),
          '| lds', sum(l.startswith('\tds_') for l in ins), '| vmem', sum(bool(re.match(r'\t(global|buffer|flat|scratch)_', l)) for l in ins))
    if '--loops' in sys.argv:
        # instruction count between loop headers (static)
        cur, n = None, 0
        for k, l in enumerate(body):
            m = re.match(r'^(\.LBB\d+_\d+):.*(Loop Header|Parent Loop|in Loop).*', l)
            if l.startswith('\t') and not l.startswith('\t.') and not l.startswith('\t;'):
                n += 1
            if re.match(r'^\.LBB\d+_\d+:.*Loop Header: Depth=(\d+)', l) or re.match(r'^\.LBB\d+_\d+:.*Parent Loop.*', l):
                print('    line', k, l.split(';')[0].strip(), l.split(';')[-1].strip(), '| instructions so far', n)
