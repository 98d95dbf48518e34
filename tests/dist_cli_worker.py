"""Launched by tests under torch.distributed.run: the kbbq command line on every rank."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'kbbq-py_amd'))

from kbbq import main   # noqa: E402

if __name__ == '__main__':
    sys.argv = ['kbbq'] + sys.argv[1:]
    main.main()
