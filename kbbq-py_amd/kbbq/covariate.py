"""
kbbq.covariate -- the covariate table classes of the reference
(reference kbbq/covariate.py:23-465) with the tallying done on the MI355X.

Covariate / RGCovariate / QCovariate / CycleCovariate / DinucCovariate keep the
reference's attributes (.errors, .total), growth rules (pad_axis appends float
zeros; CycleCovariate.pad_axis keeps the negative-cycle half at the tail) and
accessors.  CovariateData.consume_read(ReadData) implements the INTENDED
semantics -- the same tallies as recalibrate.fastq_to_covariate_arrays
(the reference's own consume_read only works for its
one-error toy fixture) -- by running the K1 kernel on the read:
    counted  : positions with skips == False (whatever their quality)
    errors   : counted positions with errors == True
    dinuc    : counted positions whose dinucleotide context is defined
               (position >= 1, q >= minscore, neither base N)
CovariateData.consume_batch(ReadBatch) is the bulk form for device-resident reads.
"""
import numpy as np

from . import compare_reads
from . import read as _read            # noqa: F401  (API parity: kbbq.covariate.read)


def pad_axis(array, axis, n):
    """Append n zeros along `axis` (the result is float64, as np.append of float zeros is)."""
    shape = array.shape[0:axis] + (n,) + array.shape[axis + 1:]
    return np.append(array, np.zeros(shape), axis=axis)


class Covariate():
    """errors / total arrays indexed by covariate value."""

    def __init__(self, shape=0):
        self.errors = np.zeros(shape, dtype=np.int_)
        self.total = np.zeros(shape, dtype=np.int_)

    def pad_axis(self, axis, n=1):
        self.errors = pad_axis(self.errors, axis=axis, n=n)
        self.total = pad_axis(self.total, axis=axis, n=n)

    def pad_axis_to_fit(self, axis, idx):
        size = self.shape()[axis]
        if idx < -size or idx >= size:
            self.pad_axis(axis=axis, n=(-idx - size) if idx < 0 else (idx - size + 1))

    def increment(self, idx, value=(1, 1)):
        np.add.at(self.errors, idx[0], value[0])
        np.add.at(self.total, idx[1], value[1])

    def shape(self):
        assert self.total.shape == self.errors.shape
        return self.total.shape

    def __getitem__(self, key):
        return (self.errors[key], self.total[key])

    def __setitem__(self, key, value):
        self.errors[key] = value[0]
        self.total[key] = value[1]

    def _add(self, errs, total):
        """Add device-produced count arrays (already padded to fit)."""
        sl = tuple(slice(0, s) for s in errs.shape)
        self.errors[sl] += errs
        self.total[sl] += total


class RGCovariate(Covariate):
    def __init__(self):
        super().__init__(shape=0)

    def consume_read(self, read):
        rge, rgv = read.get_rg_errors()
        if len(rgv):
            self.pad_axis_to_fit(axis=0, idx=rgv[0])
        self.increment((rge, rgv))
        return rge, rgv

    def num_rgs(self):
        return self.shape()[0]


class QCovariate(Covariate):
    def __init__(self):
        self.rgcov = RGCovariate()
        super().__init__(shape=(0, 0))

    def consume_read(self, read):
        rge, rgv = self.rgcov.consume_read(read)
        self.pad_axis_to_fit(axis=0, idx=self.rgcov.num_rgs() - 1)
        qe, qv = read.get_q_errors()
        if len(qv):
            self.pad_axis_to_fit(axis=1, idx=np.amax(qv))
        self.increment(idx=((rge, qe), (rgv, qv)))
        return (rge, rgv), (qe, qv)

    def num_qs(self):
        return self.shape()[1]


class CycleCovariate(Covariate):
    def __init__(self):
        super().__init__(shape=(0, 0, 0))

    def pad_axis(self, axis, n=1):
        """Growing the cycle axis keeps the first half at the front and the negative-cycle
        half at the new tail (reference covariate.py:312-341)."""
        if not (axis == 2 or axis == -1):
            return super().pad_axis(axis=axis, n=n)
        if n % 2 != 0:
            raise ValueError('n should be even for the 2nd axis of a CycleCovariate. '
                             'n = {} was given.'.format(n))
        old = self.shape()[2]
        if old == 0:
            return super().pad_axis(axis=axis, n=n)
        half = old // 2
        grown = []
        for arr in (self.errors, self.total):
            new = np.zeros(arr.shape[0:2] + (old + n,), dtype=np.int_)
            new[..., 0:half] = arr[..., 0:half]
            new[..., -half:] = arr[..., -half:]
            grown.append(new)
        self.errors, self.total = grown

    def num_cycles(self):
        return self.shape()[-1] / 2


class DinucCovariate(Covariate):
    def __init__(self):
        super().__init__(shape=(0, 0, len(compare_reads.Dinucleotide.dinucs)))

    def num_dinucs(self):
        return self.shape()[-1]


class CovariateData():
    """All covariate tables needed to recalibrate: .qcov (with .qcov.rgcov), .cyclecov, .dinuccov."""

    def __init__(self):
        self.qcov = QCovariate()
        self.cyclecov = CycleCovariate()
        self.dinuccov = DinucCovariate()

    # -- device tallies -------------------------------------------------
    def _absorb(self, pos_e, pos_t, dn_e, dn_t):
        """Fold [R,43,S2] / [R,43,16] device counts into the growable host tables."""
        nz_q = np.flatnonzero(pos_t.sum(axis=(0, 2)))
        nz_r = np.flatnonzero(pos_t.sum(axis=(1, 2)))
        if nz_q.size == 0:
            return
        R, Q, S2 = int(nz_r[-1]) + 1, int(nz_q[-1]) + 1, pos_t.shape[2]
        self.qcov.rgcov.pad_axis_to_fit(0, R - 1)
        self.qcov.pad_axis_to_fit(0, R - 1); self.qcov.pad_axis_to_fit(1, Q - 1)
        for cov in (self.cyclecov, self.dinuccov):
            cov.pad_axis_to_fit(0, self.get_num_rgs() - 1)
            cov.pad_axis_to_fit(1, self.get_num_qs() - 1)
        self.cyclecov.pad_axis_to_fit(2, S2 - 1)
        half = S2 // 2
        qe, qt = pos_e.sum(axis=2), pos_t.sum(axis=2)
        self.qcov.rgcov._add(qe.sum(axis=1)[:R], qt.sum(axis=1)[:R])
        self.qcov._add(qe[:R, :Q], qt[:R, :Q])
        # device columns: first-in-pair cycles at the front, second-in-pair from the tail
        self.cyclecov.errors[:R, :Q, :half] += pos_e[:R, :Q, :half]
        self.cyclecov.total[:R, :Q, :half] += pos_t[:R, :Q, :half]
        self.cyclecov.errors[:R, :Q, -half:] += pos_e[:R, :Q, half:]
        self.cyclecov.total[:R, :Q, -half:] += pos_t[:R, :Q, half:]
        self.dinuccov._add(dn_e[:R, :Q], dn_t[:R, :Q])

    def consume_read(self, read, minscore=6):
        """Add one ReadData to the tables (K1 kernel; see module docstring for the rules)."""
        from . import _device as dev
        L = len(read)
        if L == 0:
            return
        qual = np.asarray(read.qual, dtype=np.int64)
        if np.any(qual > 42):
            raise IndexError('quality above 42: the device Q axis is fixed at 43')
        codes = np.asarray(read.seq, dtype='U1').view(np.uint32).astype(np.int64)
        if np.any(codes > 255):
            raise TypeError('non-ASCII base')
        pitch = max(16, (L + 15) // 16 * 16)
        seq = np.full((1, pitch), ord('N'), dtype=np.uint8); seq[0, :L] = codes
        cseq = seq.copy()
        err = np.asarray(read.errors, dtype=bool)
        cseq[0, :L][err] = np.where(seq[0, :L][err] == ord('A'), ord('C'), ord('A'))
        q = np.zeros((1, pitch), dtype=np.uint8)
        q[0, :L] = np.where(np.asarray(read.skips, dtype=bool), 0, qual + 33)   # skipped: never counted
        rg = read.get_rg_int()
        meta = np.array([L | (rg << 16) | (int(bool(read.second)) << 31)], dtype=np.uint32)
        batch = dev.ReadBatch.from_host(seq, q, meta, cseq=cseq)
        tables = dev.Tables(rg + 1, 2 * L)
        dev.accumulate(batch, tables, minscore=0, dinuc_minscore=minscore)
        self._absorb(*tables.to_host())

    def consume_batch(self, batch, R, S, minscore=6):
        """Bulk form: tally a device-resident ReadBatch with the recalibrate rules
        (q < minscore is never counted), S = longest read."""
        from . import _device as dev
        tables = dev.Tables(R, 2 * S)
        dev.accumulate(batch, tables, minscore=minscore)
        self._absorb(*tables.to_host())

    # -- sizes ----------------------------------------------------------
    def get_num_rgs(self):
        return self.qcov.rgcov.num_rgs()

    def get_num_qs(self):
        return self.qcov.num_qs()

    def get_num_cycles(self):
        return self.cyclecov.num_cycles()

    def get_num_dinucs(self):
        return self.dinuccov.num_dinucs()
