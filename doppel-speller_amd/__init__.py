"""MI355X-native drop-in for doppel-speller's candidate-generation-and-scoring hot path.

Public surface (mirrors the reference, doppelspeller/match_maker.py and doppelspeller/feature_engineering.py):

    MatchMaker(data, truth_data, top_n).get_closest_matches(row_number)   -> list of title_id
    construct_features(title_number_of_characters, truth_number_of_characters, title, title_truth,
                       truth_words_counts, space_code, number_of_truth_titles, dummy, response)   (in place)
    FEATURES_COUNT, encode_title, get_truth_words_counts

All arithmetic runs in hand-written HIP kernels (csrc/*.hip -> libdoppel_amd.so, C ABI in include/doppel_amd.h);
there is no CPU fallback -- importing works without the library, calling anything that computes does not.
"""
from . import _lib  # noqa: F401
from ._lib import DoppelError, build_library, library_path  # noqa: F401
from .feature_engineering import (  # noqa: F401
    FEATURES_COUNT, TitleTable, construct_features, construct_features_indexed, encode_title, encode_titles,
    get_truth_words_counts, levenshtein_ratio_batch, find_close_matches, ALLOWED_CHARACTERS, SPACE_CODE, SORT_KEY)
from .match_maker import MatchMaker, NativeProblem, TruthIndex  # noqa: F401
from .pipeline import CandidatePipeline  # noqa: F401
from .forest import ForestModel  # noqa: F401
from .text import transform_title, transform_titles  # noqa: F401
