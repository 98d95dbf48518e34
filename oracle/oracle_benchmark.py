"""
oracle_benchmark.py -- CPU ORACLE for the benchmark-path error flagging (SURVEY.md section
8(f) #1, BASELINE.json configs[4]).  TEST INFRASTRUCTURE ONLY.

Restates, read by read and CIGAR operation by operation, the reference's
  compare_reads.find_read_errors             (kbbq/compare_reads.py:84-139)
  benchmark.get_ref_dict / get_var_sites / get_bed_dict / get_full_skips   (benchmark.py:9-39)
  benchmark.get_error_dict / calculate_q / benchmark_fastq / benchmark_bam / print_benchmark
                                             (benchmark.py:57-143)
on the TEXT formats (SAM, FASTA, VCF, BED) through the stand-in readers of oracle/_shim.py.
Pinned by the reference's own known answers on the SAM-spec example
(tests/test_compare_reads.py:87-122, tests/test_benchmark.py:7-157, restated in
tests/test_oracle_benchmark.py) and by golden vectors from running the unmodified reference
under the shim on synthetic alignments (oracle/gen_golden.py -> tests/golden/bench_*.npz).

Python negative-index behaviour is part of the restated semantics: an insertion at the first
reference position of a read looks at subset_variable[-1] (the LAST position of the read's
reference window), a deletion before the first read base ORs into skips[-1] (the LAST base).
"""
import numpy as np

import _shim
import oracle as O


def get_ref_dict(fasta_path):
    """benchmark.py:9-12 -> {chrom: uint8 array of the characters}."""
    fa = _shim.FastaFile(fasta_path)
    return {c: np.frombuffer(fa.fetch(reference=c).encode('ascii'), dtype=np.uint8) for c in fa.references}


def get_var_sites(vcf_path):
    """benchmark.py:14-20 / compare_reads.py:54-68: every 0-based position covered by a record."""
    d = {}
    for rec in _shim.VariantFile(vcf_path):
        for i in range(rec.start, rec.stop):
            d.setdefault(rec.chrom, []).append(i)
    return d


def get_full_skips(refdict, var_sites, bed_path=None):
    """benchmark.py:22-39: variant sites, plus everything outside the BED when one is given.
    KeyError when a contig has no variant (var_sites[chrom], :32) -- as in the reference."""
    skips = {c: np.zeros(len(refdict[c]), dtype=bool) for c in refdict}
    for c in skips:
        skips[c][np.array(var_sites[c], dtype=np.int64)] = True
    if bed_path is not None:
        bed = {c: np.zeros(len(refdict[c]), dtype=bool) for c in refdict}
        with open(bed_path) as fh:
            for rec in _shim.tabix_iterator(fh):
                bed[rec.contig][rec.start:rec.end] = True
        for c in skips:
            skips[c][~bed[c]] = True
    return skips


def find_read_errors(read, ref, variable):
    """compare_reads.py:84-139, one base at a time."""
    seq = np.frombuffer(read.query_sequence.encode('ascii'), dtype=np.uint8)
    n = len(seq)
    skips = np.zeros(n, dtype=bool)
    errors = np.zeros(n, dtype=bool)
    sub_var = variable[read.reference_name][read.reference_start:read.reference_end]
    refseq = ref[read.reference_name][read.reference_start:read.reference_end]
    readidx = refidx = 0
    for op, l in read.cigartuples:
        if op in (0, 7, 8):                                        # M = X   (:109-114)
            if refidx + l > len(refseq) or readidx + l > n:
                raise ValueError('operands could not be broadcast together')
            errors[readidx:readidx + l] = refseq[refidx:refidx + l] != seq[readidx:readidx + l]
            skips[readidx:readidx + l] = sub_var[refidx:refidx + l]
            readidx += l; refidx += l
        elif op == 1:                                              # I       (:115-120)
            skips[readidx:readidx + l] = bool(sub_var[refidx - 1]) and bool(sub_var[refidx])   # IndexError at the window end
            readidx += l
        elif op in (2, 3):                                         # D N     (:121-125)
            skips[readidx - 1] = skips[readidx - 1] or bool(np.any(sub_var[refidx:refidx + l]))
            refidx += l
        elif op == 4:                                              # S       (:126-129)
            skips[readidx:readidx + l] = True
            readidx += l
        elif op in (5, 6):                                         # H P     (:130-134)
            continue
        else:
            raise ValueError('Unrecognized Cigar Operation %s In Read\n%s' % (op, read))
    return errors, skips


def bam_readname(read):
    return read.query_name + ('/2' if read.is_read2 else '/1')       # benchmark.py:41-48


def fastq_readname(name):
    return name.split('_')[0]                                       # benchmark.py:50-55


def get_error_dict(reads, refdict, fullskips):
    """benchmark.py:57-74: flags flipped for reverse-strand reads (samtools fastq re-reverses them)."""
    d = {}
    for read in reads:
        e, s = find_read_errors(read, refdict, fullskips)
        if read.is_reverse:
            e, s = np.flip(e), np.flip(s)
        d[bam_readname(read)] = (e, s)
    return d


def calculate_q(errors, quals):
    """benchmark.py:76-91."""
    numtotal = np.bincount(quals.reshape(-1))
    numerrs = np.bincount(quals[errors].reshape(-1), minlength=len(numtotal))
    nz = numtotal != 0
    q = O.p_to_q(np.true_divide(numerrs[nz], numtotal[nz]))
    actual = np.zeros(len(numtotal), dtype=np.int64)
    actual[nz] = q
    return actual, numtotal


def benchmark_bam(reads, ref, var_sites, use_oq=False, bed_path=None):
    """benchmark.py:118-127."""
    fullskips = get_full_skips(ref, var_sites, bed_path)
    es, ss, qs = [], [], []
    for r in reads:
        e, s = find_read_errors(r, ref, fullskips)
        q = np.array([ord(c) - 33 for c in r.get_tag('OQ')], dtype=np.int64) if use_oq \
            else np.array(r.query_qualities, dtype=np.int64)
        es.append(e); ss.append(s); qs.append(q)
    e, s, q = np.concatenate(es), np.concatenate(ss), np.concatenate(qs)
    return calculate_q(e[~s], q[~s])


def benchmark_fastq(fastq_path, reads, ref, var_sites, bed_path=None):
    """benchmark.py:93-104: FASTQ reads joined to the BAM's flags by canonical name."""
    fullskips = get_full_skips(ref, var_sites, bed_path)
    edict = get_error_dict(reads, ref, fullskips)
    es, ss, qs = [], [], []
    for name, _, qual in O.read_fastq(fastq_path):
        e, s = edict[fastq_readname(name)]                          # KeyError when absent, as in the reference
        es.append(e); ss.append(s); qs.append(np.array([ord(c) - 33 for c in qual], dtype=np.int64))
    e, s, q = np.concatenate(es), np.concatenate(ss), np.concatenate(qs)
    return calculate_q(e[~s], q[~s])


def format_benchmark(actual_q, label, nbases):
    """benchmark.py:129-143 as one string."""
    nz = nbases != 0
    return ''.join('%d\t%d\t%s\t%d\n' % (pq, aq, label, nb)
                   for pq, aq, nb in zip(np.arange(len(actual_q))[nz], actual_q[nz], nbases[nz]))


# --------------------------------------------------------------------------
# synthetic truth set (SAM + FASTA + VCF + BED + FASTQ), deterministic in `seed`
# --------------------------------------------------------------------------
_COMP = {'A': 'T', 'C': 'G', 'G': 'C', 'T': 'A', 'N': 'N'}


def synth_truthset(outdir, seed, npairs=200, readlen=(36, 150), contigs=(('chr1', 5000), ('chr2', 3000))):
    """Writes ref.fa, aln.sam, vars.vcf, conf.bed, reads.fq under outdir and returns their paths.
    Reads carry substitutions, insertions, deletions, N-skips, soft and hard clips, both strands,
    both mates, OQ tags; reads.fq is what `samtools fastq -t -N -O` + the tutorial's `tr` would
    give (reverse-strand reads reverse-complemented, qualities reversed, name/1|/2_RG:Z:id)."""
    import os
    rng = np.random.default_rng(seed)
    bases = np.array(list('ACGT'))
    ref = {c: ''.join(rng.choice(bases, size=L)) for c, L in contigs}
    paths = {k: os.path.join(outdir, v) for k, v in dict(fa='ref.fa', sam='aln.sam', vcf='vars.vcf',
                                                          bed='conf.bed', fq='reads.fq').items()}
    with open(paths['fa'], 'w') as fh:
        for c, _ in contigs:
            fh.write('>%s\n' % c)
            for i in range(0, len(ref[c]), 60):
                fh.write(ref[c][i:i + 60] + '\n')
    with open(paths['vcf'], 'w') as fh:
        fh.write('##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n')
        for c, L in contigs:                      # every contig needs >= 1 record (benchmark.py:32)
            pos = np.sort(rng.choice(np.arange(1, L - 5), size=max(3, L // 60), replace=False))
            for p in pos:
                rl = int(rng.integers(1, 4))
                fh.write('%s\t%d\t.\t%s\t%s\t30\t.\t.\n' % (c, p + 1, ref[c][p:p + rl], 'A'))
    with open(paths['bed'], 'w') as fh:
        for c, L in contigs:
            a = 0
            while a < L:
                b = a + int(rng.integers(200, 900))
                fh.write('%s\t%d\t%d\n' % (c, a + int(rng.integers(0, 40)), min(b, L + 3)))
                a = b + int(rng.integers(0, 60))
    sam, fq = ['@HD\tVN:1.6\tSO:unsorted'] + ['@SQ\tSN:%s\tLN:%d' % c for c in contigs], []
    for i in range(npairs):
        for mate in (1, 2):
            c, L = contigs[int(rng.integers(0, len(contigs)))]
            want = int(rng.integers(readlen[0], readlen[1] + 1))
            start = int(rng.integers(1, L - 2 * readlen[1] - 50))
            ops, seq, rp = [], [], start
            if rng.random() < 0.15:
                ops.append((5, int(rng.integers(1, 10))))               # hard clip
            if rng.random() < 0.3:
                l = int(rng.integers(1, 12)); ops.append((4, l)); seq.append(''.join(rng.choice(bases, size=l)))
            if rng.random() < 0.05 and not any(o == 4 for o, _ in ops):
                l = int(rng.integers(1, 4)); ops.append((2, l)); rp += l   # leading deletion (skips[-1] case)
            qlen = sum(len(x) for x in seq)
            while qlen < want:
                l = int(min(want - qlen, rng.integers(5, 60)))
                seg = list(ref[c][rp:rp + l])
                for k in np.flatnonzero(rng.random(l) < 0.03):
                    seg[k] = str(rng.choice(bases))                  # substitution (may equal the reference)
                ops.append((int(rng.choice([0, 0, 0, 7, 8])), l)); seq.append(''.join(seg)); rp += l; qlen += l
                r = rng.random()
                if qlen < want - 1 and r < 0.2:          # never the last reference-consuming op (:118 would IndexError)
                    l = int(min(want - qlen - 1, rng.integers(1, 6))); ops.append((1, l))
                    seq.append(''.join(rng.choice(bases, size=l))); qlen += l
                elif qlen < want and r < 0.4:
                    l = int(rng.integers(1, 8)); ops.append((int(rng.choice([2, 2, 3])), l)); rp += l
            if rng.random() < 0.25:
                l = int(rng.integers(1, 10)); ops.append((4, l)); seq.append(''.join(rng.choice(bases, size=l)))
            if rng.random() < 0.1:
                ops.append((5, int(rng.integers(1, 10))))
            s = ''.join(seq)
            q = ''.join(chr(33 + int(x)) for x in rng.integers(2, 42, size=len(s)))
            oq = ''.join(chr(33 + int(x)) for x in rng.integers(2, 42, size=len(s)))
            rev = bool(rng.random() < 0.5)
            flag = 1 | (64 if mate == 1 else 128) | (16 if rev else 0)
            cigar = ''.join('%d%s' % (l, 'MIDNSHP=X'[o]) for o, l in ops)
            rg = 'g%d' % (i % 3)
            sam.append('r%05d\t%d\t%s\t%d\t60\t%s\t=\t%d\t0\t%s\t%s\tRG:Z:%s\tOQ:Z:%s'
                       % (i, flag, c, start + 1, cigar, start + 1, s, q, rg, oq))
            fs, fqq = (''.join(_COMP[b] for b in reversed(s)), oq[::-1]) if rev else (s, oq)
            fq.append('@r%05d/%d_RG:Z:%s\n%s\n+\n%s\n' % (i, mate, rg, fs, fqq))
    with open(paths['sam'], 'w') as fh:
        fh.write('\n'.join(sam) + '\n')
    with open(paths['fq'], 'w') as fh:
        fh.write(''.join(fq))
    return paths
