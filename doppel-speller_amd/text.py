"""transform_title (doppelspeller/common.py:20-47): the normalisation every title goes through before n-grams, words
and encodings are derived from it.  `transform_titles` is the batch form: the Unicode step (NFD + ASCII encoding,
common.py:25-26) stays with Python's `unicodedata` and is skipped for ASCII titles, the byte work (:26-38) runs natively
(`ds_transform_titles`, host code in libdoppel_amd.so).  The line-by-line restatement of the reference function lives
with the tests' CPU restatements, not in the product."""
import ctypes
import unicodedata

import numpy as np

from . import _lib

N_GRAMS = 3                                  # settings.py:15
MAX_CHARACTERS_ALLOWED_IN_THE_TITLE = 255    # settings.py:68


def transform_title(title):
    """common.py:20-47 for one title (warnings not reproduced): the batch form below on a list of one."""
    return transform_titles([title])[0]


def transform_titles(titles):
    """`[transform_title(t) for t in titles]`, the byte work done natively in one call."""
    encoded = [t.encode('ascii') if t.isascii() else unicodedata.normalize('NFD', t).encode('ascii', 'ignore')
               for t in titles]
    offsets = np.zeros(len(encoded) + 1, dtype=np.int64)
    np.cumsum([len(e) for e in encoded], out=offsets[1:])
    chars = np.frombuffer(b"".join(encoded), dtype=np.uint8) if offsets[-1] else np.zeros(1, dtype=np.uint8)
    out_chars = np.empty(int(offsets[-1]) + len(encoded) * N_GRAMS + 1, dtype=np.uint8)
    out_offsets = np.empty(len(encoded) + 1, dtype=np.int64)
    pointer = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    _lib.check(_lib.lib().ds_transform_titles(pointer(np.ascontiguousarray(chars)), pointer(offsets), len(encoded),
                                              MAX_CHARACTERS_ALLOWED_IN_THE_TITLE, N_GRAMS, pointer(out_chars),
                                              pointer(out_offsets)), "ds_transform_titles")
    raw = out_chars[:out_offsets[-1]].tobytes().decode('ascii')
    return [raw[out_offsets[i]:out_offsets[i + 1]] for i in range(len(encoded))]
