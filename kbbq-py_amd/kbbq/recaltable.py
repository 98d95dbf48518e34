"""
GATK recalibration-report codec: the text format that stores the model of the recalibrate
path (count tables after K1) so that a later run can skip K1.

Same class and method surface as the reference's kbbq/recaltable.py (GATKReport :9-124,
GATKTable :126-391, RecalibrationReport :393-491); `.data` is a pandas DataFrame there and
here, because that attribute IS the interface.  Parsing and rendering are done on column
arrays; nothing here needs a GPU.

Text format (reverse-engineered v1.1, reference recaltable.py:192-207):
    #:GATKReport.v1.1:<ntables>
    #:GATKTable:<ncol>:<nrow>:<fmt>:<fmt>:...:;
    #:GATKTable:<title>:<description>
    header, columns two spaces apart, every column padded to its widest cell
    rows: strings flush left, numbers flush right
tables are separated by one empty line.
"""
import numpy as np
import pandas as pd


def _kind_letter(dtype):
    k = np.dtype(dtype).kind if not isinstance(dtype, str) else 'O'
    return 'd' if k in 'iu' else 'f' if k == 'f' else 's'


class GATKReport:
    """A report = a version string and a list of tables (reference recaltable.py:9-124)."""

    def __init__(self, tables, version='1.1'):
        self.tables = tables
        self.version = version

    @classmethod
    def fromfile(cls, filename):
        """reference recaltable.py:46-66: header `#:GATKReport.v<version>:<ntables>`, tables
        separated by blank lines; a table count that disagrees with the header is a ValueError."""
        with open(filename) as fh:
            first = fh.readline()
            body = fh.read()
        _, version, ntables = first.strip().split(':')
        version = version.split(sep='v', maxsplit=1)[-1]
        tables = [GATKTable.fromstring(chunk) for chunk in body.split('\n\n') if chunk != '']
        if len(tables) != int(ntables):
            raise ValueError('Malformed or truncated file %s: the header (%s) announces %s tables, %d found.'
                             % (filename, first.strip(), ntables, len(tables)))
        return cls(tables, version)

    def get_headerstring(self):
        return '#:GATKReport.v%s:%d' % (self.version, len(self.tables))

    def write(self, filename):
        with open(filename, 'w') as fh:
            fh.write(str(self))

    def __str__(self):
        return self.get_headerstring() + '\n' + ''.join(str(t) + '\n\n' for t in self.tables)

    def __repr__(self):
        return self.get_headerstring() + '\n' + '\n'.join([repr(t) for t in self.tables]) + '\n'

    def __eq__(self, other):
        if type(other) is not type(self):
            return NotImplemented
        return (self.version == other.version and len(self.tables) == len(other.tables)
                and all(s == o for s, o in zip(self.tables, other.tables)))


class GATKTable:
    """One table: title, description, DataFrame (reference recaltable.py:126-391)."""

    def __init__(self, title, description, data):
        self.title = title
        self.description = description
        self.data = data
        # same two public dicts as the reference (:175-184); get_colfmts reads them
        self.typemap = {np.dtype(np.int_): 'd', np.dtype(np.float64): 'f', str: 's', np.dtype(object): 's'}
        self.precisionmap = {'EmpiricalQuality': '.4', 'EstimatedQReported': '.4', 'Errors': '.2'}

    @classmethod
    def fromstring(cls, tablestring):
        """reference recaltable.py:186-217: line 0 formats, line 1 title/description, line 2
        header, then whitespace-separated rows."""
        lines = tablestring.splitlines()
        title, description = lines[1].split(':')[2:4]
        header = lines[2].split()
        types = cls.parse_fmtstring(header, lines[0])
        cells = [ln.split() for ln in lines[3:]]
        cols = list(zip(*cells)) if cells else [()] * len(header)
        frame = pd.DataFrame({h: list(c) for h, c in zip(header, cols)})
        return cls(title, description, frame.astype(types))

    @staticmethod
    def parse_fmtstring(header, fmtstring):
        """{column: type} from `#:GATKTable:ncol:nrow:%d:%.4f:%s:;` (reference :219-244);
        formats ending in anything but d / f / s give no entry."""
        fmts = fmtstring.split(':')[4:-1]
        kinds = {'d': np.int64, 'f': np.float64, 's': str}
        out = {}
        for h, f in zip(header, fmts):
            if f[-1:] in kinds:
                out[h] = kinds[f[-1]]
        return out

    def get_unindexed(self):
        """Copy with a named index turned back into columns (reference :267-279)."""
        if self.data.index.names == [None]:
            return self.data.copy()
        return self.data.reset_index()

    def get_colfmts(self):
        """['%s', '%.4f', '%d', ...] from dtypes, typemap and precisionmap (reference :281-300)."""
        flat = self.get_unindexed()
        out = []
        for dt, head in zip(flat.dtypes, flat.columns.to_list()):
            letter = self.typemap[dt] if dt in self.typemap else _kind_letter(dt)
            out.append('%' + self.precisionmap.get(head, '') + letter)
        return out

    def get_fmtstring(self):
        return ':'.join(['#', 'GATKTable', str(self.get_ncols()), str(self.get_nrows())] + self.get_colfmts() + [';'])

    def get_titlestring(self):
        return ':'.join(['#', 'GATKTable', self.title, self.description])

    def get_datastring(self):
        """Header + rows (reference :316-347).  Column width = max(header, widest cell), where
        -- as in the reference, whose per-column formatter closures all end up bound to the LAST
        column's format -- the width of every numeric column is measured on its values rendered
        with the last column's format, while the cells themselves use the column's own format
        applied to float(value)."""
        fmts = self.get_colfmts()
        flat = self.get_unindexed()
        header = flat.columns.to_list()
        nrows = flat.shape[0]
        cols = [flat[h].to_numpy() for h in header]
        widths = [len(h) for h in header]
        if nrows:
            last = fmts[-1]
            for i, (f, col) in enumerate(zip(fmts, cols)):
                w = max(len(x) for x in col) if f == '%s' else max(len(last % x) for x in col.tolist())
                widths[i] = max(widths[i], w)
        lines = ['  '.join(h.ljust(w) for h, w in zip(header, widths))]
        rendered = []
        for f, col, w in zip(fmts, cols, widths):
            if f == '%s':
                rendered.append([x.ljust(w) for x in col])
            else:
                rendered.append([(f % float(x)).rjust(w) for x in col.tolist()])
        lines.extend('  '.join(row) for row in zip(*rendered))
        return '\n'.join(lines)

    def get_nrows(self):
        return self.get_unindexed().shape[0]

    def get_ncols(self):
        return self.get_unindexed().shape[1]

    def write(self, filehandle):
        return filehandle.write(str(self) + '\n')

    def __str__(self):
        return self.get_fmtstring() + '\n' + self.get_titlestring() + '\n' + self.get_datastring()

    def __repr__(self):
        return self.get_fmtstring() + '\n' + self.get_titlestring() + '\n' + repr(self.data)

    def __eq__(self, other):
        if type(other) is not type(self):
            return NotImplemented
        return (self.title == other.title and self.description == other.description
                and self.data.equals(other.data))


class RecalibrationReport(GATKReport):
    """The five-table BQSR report (reference recaltable.py:393-491): Arguments, Quantized,
    RecalTable0 (read group), RecalTable1 (+ reported quality), RecalTable2 (+ covariate).
    Construction sets the indices and dtypes the model code relies on."""

    TITLES = ('Arguments', 'Quantized', 'RecalTable0', 'RecalTable1', 'RecalTable2')

    def __init__(self, tables, version='1.1'):
        super().__init__(tables, version)
        if len(self.tables) != 5:
            raise ValueError('A RecalibrationReport should have 5 tables. This report contains %d.'
                             % len(self.tables))
        for t, title in zip(self.tables, self.TITLES):
            assert t.title == title
        t = self.tables
        t[0].data = t[0].data.set_index('Argument')
        t[1].data = t[1].data.astype({'QualityScore': np.int_, 'Count': np.longlong,
                                      'QuantizedScore': np.int_}).set_index('QualityScore')
        t[2].data = t[2].data.set_index('ReadGroup')
        t[3].data = t[3].data.astype({'ReadGroup': str, 'QualityScore': np.int_}) \
                             .set_index(['ReadGroup', 'QualityScore'])
        t[4].data = t[4].data.astype({'ReadGroup': str, 'QualityScore': np.int_,
                                      'CovariateName': str, 'CovariateValue': str}) \
                             .set_index(['ReadGroup', 'QualityScore', 'CovariateName', 'CovariateValue'])

    def __str__(self):
        """GATK prints CovariateValue before CovariateName; the index keeps them the other way
        round (reference :479-491)."""
        held = self.tables[4].data
        self.tables[4].data = held.swaplevel('CovariateValue', 'CovariateName')
        try:
            return super().__str__()
        finally:
            self.tables[4].data = held
