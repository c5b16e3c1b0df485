#!/usr/bin/env python3
"""Golden vectors for `transform_title` (doppelspeller/common.py:20-47), captured by running the reference's own
function in the build container (same stand-ins for `numba` / `Levenshtein` as make_golden.py; nothing is written
under /root/reference).  Usage:  python tests/golden/make_golden_transform.py"""
import gzip
import json
import logging
import os
import random
import sys

sys.dont_write_bytecode = True
os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden  # noqa: E402  (only for its shims; its main() is not run)

make_golden._install_shims()
sys.path.insert(0, make_golden.REFERENCE)
os.environ.setdefault("PROJECT_DATA_PATH", "/tmp")
logging.disable(logging.CRITICAL)
from doppelspeller import common  # noqa: E402

hand_made = [
    "", " ", "A", "ab", "a-b", "AbC", "  leading and trailing  ", "multiple   spaces    here", "tab\tseparated\twords",
    "new\nline", "tabs \t and  spaces", "Ünïcode-Näme B.V.", "Société Générale S.A.", "Łódź Sp. z o.o.", "ÆON Ltd",
    "straße & söhne", "naïve café — résumé", "日本語 株式会社", "ＦＵＬＬＷＩＤＴＨ ltd", "①②③ numbers", "x" * 300,
    ("word " * 80), "a" * 254 + " b", "a" * 255, "a" * 256, " " * 10 + "z", "!!!", "@#$%", "1-2-3", "--", "o'neil & sons (uk) ltd.",
    "ß", "ǅ", "İstanbul A.Ş.", "Ελληνικά ΑΕ", "mixed nbsp", "zero​width", "é combining", "3M", "  -  ",
]
rng = random.Random(7)
with gzip.open(f"{make_golden.REFERENCE}/example_dataset/example_truth.csv.gz", "rt", encoding="utf-8") as handle:
    lines = handle.read().splitlines()[1:]
sampled = [line.split("|", 1)[-1] if "|" in line else line.split(",", 1)[-1] for line in rng.sample(lines, 300)]
titles = hand_made + sampled
vectors = [{"title": t, "transformed": common.transform_title(t)} for t in titles]
with open(os.path.join(HERE, "transform_title.json"), "w", encoding="utf-8") as out:
    json.dump(vectors, out, ensure_ascii=True, indent=0)
print(len(vectors), "vectors;", sum(1 for v in vectors if not v["title"].isascii()), "with non-ASCII input")

# ---- the three vectors the reference's OWN tests hold (doppelspeller/tests/test_common.py:16-28), captured as data: the inputs
# are those of its test case, the outputs what its functions return here (asserted equal to the values its tests expect).
import pandas as pd  # noqa: E402
from doppelspeller import constants as c  # noqa: E402

kat_title = '''LKJblksd skjasl dfkjf &* 8*&&&8 GGdjsdkj--sdsd-"sdi..//' d'  k   bkjh77_asda33'''
ground_truth = [['first', 'second', 'first', 'third', 'first'], ['first', 'first'], ['fifth']]
frame = pd.DataFrame(index=range(len(ground_truth)))
frame.loc[:, c.COLUMN_WORDS] = ground_truth
counter = common.get_words_counter(frame)
reference_tests = {
    "transform_title": {"title": kat_title, "transformed": common.transform_title(kat_title)},
    "words_counter": {"titles_as_words": ground_truth, "counts": dict(counter)},
    "idf_word": {"word": "first", "number_of_titles": len(ground_truth), "count": counter["first"],
                 "idf": common.idf_word("first", counter, len(ground_truth))},
}
assert reference_tests["transform_title"]["transformed"] == 'lkjblksd skjasl dfkjf 88 ggdjsdkj sdsd sdi d k bkjh77asda33'
assert reference_tests["words_counter"]["counts"] == {'first': 2, 'second': 1, 'third': 1, 'fifth': 1}
assert round(reference_tests["idf_word"]["idf"], 5) == 0.40547
with open(os.path.join(HERE, "reference_tests.json"), "w", encoding="utf-8") as out:
    json.dump(reference_tests, out, ensure_ascii=True, indent=1)
print("reference_tests.json:", reference_tests["idf_word"])
