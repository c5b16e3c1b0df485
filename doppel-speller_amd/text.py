"""transform_title (doppelspeller/common.py:20-47): the normalisation every title goes through before n-grams, words
and encodings are derived from it.  `transform_title` is the reference's function restated; `transform_titles` is the
batch form: the Unicode step (NFD + ASCII encoding) stays with Python's `unicodedata` and is skipped for ASCII titles,
the byte work runs natively (`ds_transform_titles`, host code in libdoppel_amd.so)."""
import ctypes
import re
import unicodedata

import numpy as np

from . import _lib

N_GRAMS = 3                                  # settings.py:15
MAX_CHARACTERS_ALLOWED_IN_THE_TITLE = 255    # settings.py:68
_SUBSTITUTE_REGEX = re.compile(r' +')        # common.py:16
_KEEP_REGEX = re.compile(r'[a-zA-Z0-9\s]')   # common.py:17


def transform_title(title):
    """Transforms a title in to alpha-numeric-only (plus spaces) text (common.py:20-47, warnings not reproduced)."""
    text = unicodedata.normalize('NFD', title)                                           # :25
    text = text.encode('ascii', 'ignore').decode('utf-8').lower().replace('-', ' ')      # :26
    text = ''.join(_KEEP_REGEX.findall(text))                                            # :28
    text = _SUBSTITUTE_REGEX.sub(' ', text).strip()                                      # :30
    number_of_characters = len(text)
    text = text[:MAX_CHARACTERS_ALLOWED_IN_THE_TITLE].strip()                            # :32
    if number_of_characters < N_GRAMS:
        return text.rjust(N_GRAMS, '0')                                                  # :38
    return text


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
