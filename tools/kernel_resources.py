#!/usr/bin/env python3
"""tools/kernel_resources.py FILE [pattern]: registers, spills, scratch per kernel from a -Rpass-analysis=kernel-resource-usage log (tools/kernel_resources.sh writes build/res.txt)."""
import re, sys
pat = sys.argv[2] if len(sys.argv) > 2 else ''
cur = None
rows = {}
for l in open(sys.argv[1]):
    m = re.search(r'remark:\s+Function Name: (\S+)', l)
    if m:
        cur = m.group(1); rows[cur] = {}; continue
    m = re.search(r'remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+) \[-Rpass', l)
    if m and cur:
        rows[cur][m.group(1).replace(' ', '')] = int(m.group(2))
for k, v in rows.items():
    if pat in k:
        print(k[:86], ' '.join('%s=%s' % (a, b) for a, b in v.items() if a in ('VGPRs', 'AGPRs', 'ScratchSize', 'SGPRsSpill', 'VGPRsSpill', 'Occupancy')))
