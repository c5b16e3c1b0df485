"""Property check of the native batch transform against the pinned Python restatement on arbitrary text."""
from hypothesis import given, settings, strategies as st


@settings(max_examples=150, deadline=None)
@given(st.lists(st.text(alphabet=st.characters(blacklist_categories=("Cs",)), max_size=320), max_size=12))
def test_native_batch_equals_python_restatement(titles):
    import doppel_speller_amd as ds
    from oracle import oracle
    assert ds.transform_titles(titles) == [oracle.transform_title(t) for t in titles]
