"""Test infrastructure: SAM text -> BAM bytes (SAM specification, sections 4.1 / 4.2), so that the reader's BAM
path can be checked against its SAM path without htslib / samtools.  BGZF blocks of at most 0xff00 input bytes,
every member with the 'BC' extra field, followed by the standard empty EOF block."""
import struct
import zlib

SEQ = {c: i for i, c in enumerate('=ACMGRSVTWYHKDBN')}
OPS = {c: i for i, c in enumerate('MIDNSHP=X')}


def _reg2bin(beg, end):
    end -= 1
    for shift, base in ((14, 4681), (17, 585), (20, 73), (23, 9), (26, 1)):
        if beg >> shift == end >> shift:
            return base + (beg >> shift)
    return 0


def _tag(field):
    tag, typ, val = field.split(':', 2)
    t = tag.encode()
    if typ == 'A':
        return t + b'A' + val.encode()
    if typ == 'i':
        v = int(val)
        for code, fmt, lo, hi in (('c', '<b', -128, 127), ('C', '<B', 0, 255), ('s', '<h', -32768, 32767), ('S', '<H', 0, 65535),
                                  ('i', '<i', -2 ** 31, 2 ** 31 - 1), ('I', '<I', 0, 2 ** 32 - 1)):
            if lo <= v <= hi:
                return t + code.encode() + struct.pack(fmt, v)
        raise ValueError(val)
    if typ == 'f':
        return t + b'f' + struct.pack('<f', float(val))
    if typ in 'ZH':
        return t + typ.encode() + val.encode('latin-1') + b'\0'
    if typ == 'B':
        sub, *vals = val.split(',')
        fmt = {'c': 'b', 'C': 'B', 's': 'h', 'S': 'H', 'i': 'i', 'I': 'I', 'f': 'f'}[sub]
        conv = float if sub == 'f' else int
        return t + b'B' + sub.encode() + struct.pack('<i', len(vals)) + b''.join(struct.pack('<' + fmt, conv(v)) for v in vals)
    raise ValueError(field)


def sam_to_bam_bytes(text, header_text=True):
    lines = [ln for ln in text.split('\n') if ln.strip()]
    header = [ln for ln in lines if ln.startswith('@')]
    refs = []
    for ln in header:
        if ln.startswith('@SQ'):
            f = dict(x.split(':', 1) for x in ln.split('\t')[1:])
            refs.append((f['SN'], int(f['LN'])))
    index = {name: i for i, (name, _) in enumerate(refs)}
    htext = ('\n'.join(header) + '\n').encode('latin-1') if header and header_text else b''
    out = [b'BAM\1', struct.pack('<i', len(htext)), htext, struct.pack('<i', len(refs))]
    for name, ln in refs:
        out += [struct.pack('<i', len(name) + 1), name.encode() + b'\0', struct.pack('<i', ln)]
    for ln in lines:
        if ln.startswith('@'):
            continue
        f = ln.split('\t')
        qname, flag, rname, pos, mapq, cigar, rnext, pnext, tlen, seq, qual = f[:11]
        ref_id = -1 if rname == '*' else index[rname]
        next_id = -1 if rnext == '*' else (ref_id if rnext == '=' else index[rnext])
        ops, num = [], ''
        if cigar != '*':
            for c in cigar:
                if c.isdigit():
                    num += c
                else:
                    ops.append((int(num) << 4) | OPS[c]); num = ''
        span = sum(o >> 4 for o in ops if (o & 15) in (0, 2, 3, 7, 8))
        l_seq = 0 if seq == '*' else len(seq)
        packed = bytearray((l_seq + 1) // 2)
        for i in range(l_seq):
            packed[i >> 1] |= SEQ[seq[i]] << (4 if i % 2 == 0 else 0)
        q = bytes([0xFF] * l_seq) if qual == '*' else bytes(ord(c) - 33 for c in qual)
        body = struct.pack('<iiBBHHHiiii', ref_id, int(pos) - 1, len(qname) + 1, int(mapq),
                           _reg2bin(int(pos) - 1, int(pos) - 1 + max(span, 1)), len(ops), int(flag), l_seq, next_id, int(pnext) - 1, int(tlen))
        body += qname.encode('latin-1') + b'\0' + b''.join(struct.pack('<I', o) for o in ops) + bytes(packed) + q
        body += b''.join(_tag(x) for x in f[11:] if x)
        out += [struct.pack('<i', len(body)), body]
    return b''.join(out)


def bgzf(data, block=0xff00):
    out = []
    for at in list(range(0, len(data), block)) + [None]:
        chunk = b'' if at is None else data[at:at + block]
        if at is not None and not chunk:
            continue
        c = zlib.compressobj(6, zlib.DEFLATED, -15)
        comp = c.compress(chunk) + c.flush()
        bsize = 18 + len(comp) + 8 - 1
        out.append(b'\x1f\x8b\x08\x04' + b'\0\0\0\0' + b'\0\xff' + struct.pack('<H', 6) + b'BC' + struct.pack('<HH', 2, bsize)
                   + comp + struct.pack('<II', zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk)))
    return b''.join(out)


def write_bam(path, sam_text, **kw):
    with open(path, 'wb') as fh:
        fh.write(bgzf(sam_to_bam_bytes(sam_text, **kw)))
    return str(path)
