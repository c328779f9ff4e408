#!/usr/bin/env python3
"""Category summary of a bench.py --dump-launches table (one training step): ms, launches, TFLOP/s per family.
    python tools/launch_summary.py gpurun_out/launches.txt"""
import collections
import re
import sys


def cat(k, n):
    if k == 'wgrad':
        if '.gb.' in n: return 'wgrad gb'
        if n.startswith('disc'): return 'wgrad D'
        if 'shared' in n: return 'wgrad shared'
        return 'wgrad G other'
    if k == 'conv':
        if 'gamma|beta' in n: return 'conv gb fwd'
        if '.gb.dgrad' in n: return 'conv gb dgrad'
        if n.startswith('disc'): return 'conv D ' + ('dgrad' if 'dgrad' in n else 'fwd')
        if 'vgg' in n: return 'conv vgg ' + ('dgrad' if 'dgrad' in n else 'fwd')
        if 'shared' in n: return 'conv shared' + (' dgrad' if 'dgrad' in n else '')
        if re.match(r'(up_|head|G_middle|conv_)', n): return 'conv G resblk ' + ('dgrad' if 'dgrad' in n else 'fwd')
        return 'conv tocg/other'
    return k


def main(path):
    rows = []
    for l in open(path):
        m = re.match(r'(\S+)\s+(.*?)\s+([\d.]+) ms\s+([\d.]+) TFLOP/s\s+([\d.]+) GB/s', l)
        if m:
            rows.append((m.group(1), m.group(2), float(m.group(3)), float(m.group(4)), float(m.group(5))))
    agg = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0])
    for k, n, ms, tf, gb in rows:
        a = agg[cat(k, n)]
        a[0] += 1; a[1] += ms; a[2] += tf * ms; a[3] += gb * ms
    for c, (n, ms, fl, by) in sorted(agg.items(), key=lambda x: -x[1][1]):
        print(f'{c:24s} {n:4d} {ms:8.2f} ms  {fl / ms if ms else 0:7.1f} TF/s  {by / ms if ms else 0:8.1f} GB/s')
    print(f'{"total":24s} {len(rows):4d} {sum(r[2] for r in rows):8.2f} ms')


if __name__ == "__main__":
    main(sys.argv[1])
