"""
_shim.py -- lets the UNMODIFIED reference (/root/reference/kbbq) import in the
build container.  TEST INFRASTRUCTURE ONLY; used by oracle/gen_golden.py and
nothing else.  It contains no reference code.

Why it is needed: the reference imports pysam, khmer and
seaborn, none of which are installed, and uses NumPy aliases that NumPy 2.x
removed (np.int, np.bool, np.float, np.object, np.unicode, np.NINF).  The
reference only touches pysam on this path through FastxFile iteration
(recalibrate.py:56,141) and FastxRecord attributes (.name, .sequence, .quality,
.get_quality_array()), so a few lines of stand-in reader are enough.  The
reference's arithmetic (NumPy / SciPy calls) runs untouched.
"""
import sys
import types

import numpy as np


def install(reference_root='/root/reference'):
    for alias, target in (('int', int), ('float', float), ('bool', bool), ('object', object),
                          ('unicode', np.str_), ('NINF', -np.inf)):
        if not hasattr(np, alias):
            setattr(np, alias, target)

    class FastxRecord:
        def __init__(self, name=None, sequence=None, quality=None, comment=None):
            self.name, self.sequence, self.quality, self.comment = name, sequence, quality, comment

        def get_quality_array(self, offset=33):
            return [ord(c) - offset for c in self.quality]

        def __str__(self):
            return '@%s\n%s\n+\n%s' % (self.name, self.sequence, self.quality)

    class FastxFile:
        def __init__(self, path):
            self._fh = open(path, 'r')

        def __enter__(self):
            return self

        def __exit__(self, *a):
            self._fh.close()

        def __iter__(self):
            return self

        def __next__(self):
            head = self._fh.readline()
            if not head:
                raise StopIteration
            seq = self._fh.readline().rstrip('\n')
            self._fh.readline()
            qual = self._fh.readline().rstrip('\n')
            fields = head.rstrip('\n')[1:].split(None, 1)
            name = fields[0] if fields else ''
            comment = fields[1] if len(fields) > 1 else None
            return FastxRecord(name, seq, qual, comment)

    pysam = types.ModuleType('pysam')
    pysam.FastxRecord = FastxRecord
    pysam.FastxFile = FastxFile
    sys.modules['pysam'] = pysam
    sys.modules['khmer'] = types.ModuleType('khmer')
    sys.modules['seaborn'] = types.ModuleType('seaborn')
    sys.dont_write_bytecode = True
    if reference_root not in sys.path:
        sys.path.insert(0, reference_root)
