"""`natsort.natsorted` as the reference uses it: file names in natural (numeric-aware) order -- data.py:18,33,
run.py:322,346."""
import re


def _key(s):
    return [int(t) if t.isdigit() else t.lower() for t in re.split(r'(\d+)', str(s))]


def natsorted(seq, key=None, reverse=False):
    return sorted(seq, key=(lambda v: _key(key(v))) if key else _key, reverse=reverse)
