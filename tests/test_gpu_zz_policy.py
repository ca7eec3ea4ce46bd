"""Which kernel form the sampler picks (selection policy, not parity).  The file name sorts after every parity file: under
`pytest -x` a policy expectation that no longer holds cannot hide parity tests behind it."""
import pytest

pytestmark = pytest.mark.gpu


def test_16_bit_rows_are_chosen_by_document_and_n_kw_size():
    """rows16=None: on wherever every document is below 2^16 tokens (the four-wave form: faster than int32 rows at every size); with
    longer documents (three waves) only when n_kw is at least GibbsSampler.ROWS16_MIN_BYTES"""
    import torch
    from lda_thesis_amd.corpus import synthetic_corpus_blocks
    from lda_thesis_amd.sampler import GibbsSampler
    for zipf, V, long_doc, want in ((0.0, 1_000_000, False, True), (1.0, 100_000, False, True), (0.0, 20_000, False, True),
                                    (0.0, 20_000, True, False), (1.0, 100_000, True, True)):
        off, w, f, z = synthetic_corpus_blocks(0, 8000, 300, V, 512, 1234, "cuda", zipf_s=zipf, block=4000)
        if long_doc:
            f = f.clone()
            f[0] = 70000                                                  # one document of 70 299 tokens
        s = GibbsSampler(off, w, f, z, 512, V, 0.1, 0.01, labs=None, seed=1)
        assert (s.n_kw16 is not None) == want, (zipf, V, long_doc)
        # K = 512 with every document below 2^16 tokens: four documents per wavefront, flags per word from the library instead of
        # per-site row starts
        assert s.quad == (want and not long_doc)
        assert (s.site_row is not None) == (want and long_doc)
        if want:
            assert (0 < s.max_doc_tokens < 65536) == (not long_doc)
        s.sweep()
        s.check_status()
        del s
