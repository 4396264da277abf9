"""End to end: a BAM file on disk -> bin/te_count's code path (native BGZF/BAM decode, unit assembly,
H2D, count kernel, TSV) on one GPU.  Prints one JSON line with the wall time per stage.

    python tools/bench_e2e.py [--records 8000000] [--layout pe|se]

The BAM is synthetic: hg38-like annotation and units as in bench.py, one fixed-width record per read
with an Illumina-style name, a 100M CIGAR, random sequence and quality bytes (the hard case for the
inflater) and four small tags; written with numpy + zlib level 1 in 0xff00-byte BGZF blocks."""
import argparse
import json
import os
import struct
import sys
import tempfile
import time
import zlib

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def write_bam_fast(path, ref_names, tid, pos, flag, mapq, seed=1, pair_names=False):
    n = len(tid)
    rng = np.random.default_rng(seed)
    name_len = 28
    rec = np.zeros(n, dtype=[('bs', '<u4'), ('tid', '<i4'), ('pos', '<i4'), ('l_name', 'u1'), ('mapq', 'u1'), ('bin', '<u2'),
                             ('n_cig', '<u2'), ('flag', '<u2'), ('l_seq', '<u4'), ('ntid', '<i4'), ('npos', '<i4'), ('tlen', '<i4'),
                             ('name', 'S%d' % name_len), ('cigar', '<u4'), ('seq', 'u1', 50), ('qual', 'u1', 100), ('aux', 'S16')])
    rec['bs'] = rec.dtype.itemsize - 4
    rec['tid'], rec['pos'], rec['flag'], rec['mapq'] = tid, pos, flag, mapq
    rec['l_name'] = name_len
    rec['n_cig'] = 1
    rec['l_seq'] = 100
    rec['ntid'] = rec['npos'] = -1
    ordinal = (np.arange(n) // 2) if pair_names else np.arange(n)
    names = np.char.add(np.char.add('A00123:45:HXX:1:', np.char.zfill(ordinal.astype(str), 10)), '/')
    names = np.char.add(names, np.where(np.arange(n) % 2 == 0, '1', '2')) if pair_names else np.char.add(names, '1')
    rec['name'] = names.astype('S%d' % name_len)  # NUL padded: 27 chars + NUL
    rec['cigar'] = (100 << 4) | 0
    rec['seq'] = rng.integers(0, 256, (n, 50), dtype=np.uint8)
    rec['qual'] = rng.integers(2, 41, (n, 100), dtype=np.uint8)
    rec['aux'] = b'NHC\x01HIC\x01ASC\x62nMC\x00'
    text = ''.join('@SQ\tSN:%s\tLN:%d\n' % (r, 300000000) for r in ref_names).encode()
    head = b'BAM\x01' + struct.pack('<I', len(text)) + text + struct.pack('<I', len(ref_names))
    for r in ref_names:
        nm = r.encode() + b'\0'
        head += struct.pack('<I', len(nm)) + nm + struct.pack('<I', 300000000)
    payload = head + rec.tobytes()
    with open(path, 'wb') as fh:
        for off in range(0, len(payload), 0xff00):
            data = payload[off:off + 0xff00]
            co = zlib.compressobj(1, zlib.DEFLATED, -15)
            c = co.compress(data) + co.flush()
            fh.write(b'\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00' + struct.pack('<H', len(c) + 25) + c +
                     struct.pack('<II', zlib.crc32(data) & 0xffffffff, len(data)))
        fh.write(bytes.fromhex('1f8b08040000000000ff0600424302001b0003000000000000000000'))
    return rec.dtype.itemsize


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--records', type=int, default=8_000_000)
    ap.add_argument('--layout', default='pe', choices=['pe', 'se'])
    ap.add_argument('--scale', type=float, default=1.0)
    args = ap.parse_args()
    from te_counter_amd import synth
    paired = args.layout == 'pe'
    ann = synth.make_annotation('hg38', scale=args.scale)
    n_units = args.records // 2 if paired else args.records
    chrom, p1, p2, flag, mapq = synth.make_units(ann, n_units, seed=4242, paired=paired)
    if paired:  # two records per unit: starts p1 / p2, the unit's flags on the first mate
        tid = np.repeat(chrom, 2)
        pos = np.stack([p1, p2], 1).reshape(-1)
        fl = np.stack([flag, np.zeros_like(flag)], 1).reshape(-1)
        mq = np.repeat(mapq, 2)
    else:     # end = pos + 100 through the 100M CIGAR (the generator's spliced reads are ignored here)
        tid, pos, fl, mq = chrom, p1, flag, mapq
    refs = ann.chrom_names + ['chrUn_1']
    tid = np.where(tid < 0, len(refs) - 1, tid).astype(np.int32)
    out = {}
    with tempfile.TemporaryDirectory() as td:
        bam_path, idx_path, tsv = os.path.join(td, 'in.bam'), os.path.join(td, 'ann.tecidx'), os.path.join(td, 'out.tsv')
        t = time.time()
        rec_bytes = write_bam_fast(bam_path, refs, tid, pos.astype(np.int32), fl.astype(np.uint16), mq.astype(np.uint8),
                                   pair_names=paired)
        out['write_bam_s'] = round(time.time() - t, 2)
        out['bam_MB'] = round(os.path.getsize(bam_path) / 1e6, 1)
        out['inflated_MB'] = round(rec_bytes * len(tid) / 1e6, 1)
        ann.flat.save(idx_path)
        from te_counter_amd.measure import measureTE
        for rep in range(2):  # the first run pays library loading
            t0 = time.time()
            mte = measureTE('bench_e2e', 20)
            mte.bind_genome(idx_path)
            mte.load_genome()                                  # .tecidx -> host arrays -> fine-bin index in HBM
            t1 = time.time()
            result = (mte.parse_bampe if paired else mte.parse_bamse)(bam_path)   # decode, unit assembly, H2D, kernel
            t2 = time.time()
            mte.save_result_bulk(result, tsv)
            t3 = time.time()
            out['run%d_s' % rep] = {'load_index': round(t1 - t0, 2), 'decode_and_count': round(t2 - t1, 2),
                                    'write_tsv': round(t3 - t2, 2), 'total': round(t3 - t0, 2)}
        out['tsv_lines'] = sum(1 for _ in open(tsv))
    out['records'] = int(len(tid))
    out['Mrecords_per_s_decode_and_count'] = round(len(tid) / out['run1_s']['decode_and_count'] / 1e6, 2)
    out['Mrecords_per_s_end_to_end'] = round(len(tid) / out['run1_s']['total'] / 1e6, 2)
    out['host_cores'] = os.cpu_count()
    print(json.dumps(out))


if __name__ == '__main__':
    main()
