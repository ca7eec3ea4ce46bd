"""Import shim for the reference  --  TEST INFRASTRUCTURE, runs only in the build container.

``/root/reference/LabeledLDA.py:1-2`` and ``CascadeLDA.py:1-2`` import gensim, which is not
installed (and there is no network).  This module registers minimal stand-in modules for
``gensim.parsing.preprocessing`` and ``gensim.corpora.dictionary`` in ``sys.modules`` so that the
reference imports and runs UNCHANGED from where it lies.  The sampler itself touches only numpy;
tokenisation/stemming/dictionary pruning (gensim==2.3.0, requirements.txt:5) is third-party code that
is absent here, so parity at THAT boundary is unpinned -- the sampler is fed identical integer arrays
on both sides, so sampler parity is unaffected.

Nothing under /root/reference is copied; this file only makes ``import LabeledLDA`` possible.
It does not exist as far as the GPU box is concerned (nothing there imports it).
"""
import sys
import types

REFERENCE_DIR = "/root/reference"


def _tokenize(text):
    # the FIXTURE tokenizer (unstemmed): the committed tokenised fixtures were built with it, and re-running
    # gen_golden.py must reproduce them array for array.  The product's load_corpus stems (text.preprocess_documents);
    # which tokens the sampler is fed does not matter for sampler parity -- both sides get the same integer arrays.
    from lda_thesis_amd.text import simple_preprocess
    return simple_preprocess(text, stem=False)


def install():
    if "gensim" in sys.modules and getattr(sys.modules["gensim"], "__llda_stub__", False):
        return
    from lda_thesis_amd.text import Dictionary

    gensim = types.ModuleType("gensim")
    gensim.__llda_stub__ = True
    parsing = types.ModuleType("gensim.parsing")
    prep = types.ModuleType("gensim.parsing.preprocessing")
    corpora = types.ModuleType("gensim.corpora")
    dictionary = types.ModuleType("gensim.corpora.dictionary")

    prep.preprocess_documents = lambda docs: [_tokenize(d) for d in docs]
    dictionary.Dictionary = Dictionary
    gensim.parsing = parsing
    parsing.preprocessing = prep
    gensim.corpora = corpora
    corpora.dictionary = dictionary
    corpora.Dictionary = Dictionary
    for name, mod in (("gensim", gensim), ("gensim.parsing", parsing),
                      ("gensim.parsing.preprocessing", prep), ("gensim.corpora", corpora),
                      ("gensim.corpora.dictionary", dictionary)):
        sys.modules[name] = mod
    if REFERENCE_DIR not in sys.path:
        sys.path.insert(0, REFERENCE_DIR)


def import_reference():
    """-> (LabeledLDA module, CascadeLDA module) imported from /root/reference, unmodified."""
    import importlib.util
    import os
    install()
    mods = []
    for name in ("LabeledLDA", "CascadeLDA"):
        path = os.path.join(REFERENCE_DIR, name + ".py")
        spec = importlib.util.spec_from_file_location("_reference_" + name, path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        mods.append(mod)
    return tuple(mods)
