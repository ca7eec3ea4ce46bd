"""Host-side text utilities: tokenizer and bag-of-words dictionary.

The reference delegates both to gensim 2.3.0 (``gensim.parsing.preprocessing.preprocess_documents``
at /root/reference/LabeledLDA.py:45, CascadeLDA.py:48 and ``gensim.corpora.dictionary.Dictionary``
at LabeledLDA.py:282-284, CascadeLDA.py:451-453).  gensim is a third-party dependency that is not
vendored in the reference and not installed in this image, so this module is the build's OWN
implementation of the same interface: parity at this boundary is unpinned.
It sits upstream of the sampler: the sampler is fed integer arrays, identical on the CPU-oracle and
the GPU side, so sampler parity does not depend on it.

Interface kept (the parts the reference touches):
  Dictionary(docs)             token2id, id2token, dfs, num_docs, values(), keys(), __len__,
                               __getitem__, doc2bow(doc) -> sorted [(id, count)],
                               filter_extremes(no_below, no_above, keep_n)
  preprocess_documents(texts)  list[str] -> list[list[str]], gensim's DEFAULT_FILTERS in their order: lower-case,
                               strip tags, strip punctuation, collapse white space, strip digits ('abc123def' ->
                               'abcdef'), drop stop words, drop tokens shorter than 3, Porter-stem -- the stop-word
                               list and the stemmer are this module's own
  simple_preprocess(text)      the UNSTEMMED tokenizer the committed fixtures (tests/golden/abstracts_d3*.npz,
                               cascade_abstracts.npz, chain_quality.npz) were built with; kept so that
                               oracle/gen_golden.py reproduces them array for array
"""
import re
import string
from collections import Counter

# A compact English stop-word list (function words only).
STOPWORDS = frozenset("""
a about above across after afterwards again against all almost alone along already also although
always am among amongst an and another any anyhow anyone anything anyway anywhere are around as at
back be became because become becomes becoming been before beforehand behind being below beside
besides between beyond both but by can cannot could did do does doing done down due during each
either else elsewhere enough etc even ever every everyone everything everywhere except few for
former formerly from further had has have having he hence her here hereafter hereby herein hereupon
hers herself him himself his how however i ie if in indeed into is it its itself just last latter
latterly least less many may me meanwhile might more moreover most mostly much must my myself namely
neither never nevertheless next no nobody none nor not nothing now nowhere of off often on once one
only onto or other others otherwise our ours ourselves out over own per perhaps rather same seem
seemed seeming seems several she should since so some somehow someone something sometime sometimes
somewhere still such than that the their theirs them themselves then thence there thereafter thereby
therefore therein thereupon these they this those though through throughout thru thus to together
too toward towards under until up upon us using very via was we well were what whatever when whence
whenever where whereafter whereas whereby wherein whereupon wherever whether which while whither who
whoever whole whom whose why will with within without would yet you your yours yourself yourselves
""".split())

_WORD = re.compile(r"[a-z]+")
_TAGS = re.compile(r"<[^>]*>")


def simple_preprocess(text, min_len=3, stem=False):
    """lower-case, drop markup/punctuation/digits, drop stop words and tokens shorter than min_len."""
    text = _TAGS.sub(" ", text.lower())
    toks = [t for t in _WORD.findall(text) if len(t) >= min_len and t not in STOPWORDS]
    if stem:
        toks = [porter_stem(t) for t in toks]
    return toks


_RE_TAGS = re.compile(r"<([^>]+)>")
_RE_PUNCT = re.compile("([%s])+" % re.escape(string.punctuation))
_RE_WS = re.compile(r"(\s)+")
_RE_NUMERIC = re.compile(r"[0-9]+")


def preprocess_string(text, stem=True, min_len=3):
    """one document through gensim's DEFAULT_FILTERS, in gensim's order (gensim.parsing.preprocessing, 2.3.0:
    lower, strip_tags, strip_punctuation, strip_multiple_whitespaces, strip_numeric, remove_stopwords,
    strip_short, stem_text)."""
    s = text.lower()
    s = _RE_TAGS.sub("", s)
    s = _RE_PUNCT.sub(" ", s)
    s = _RE_WS.sub(" ", s)
    s = _RE_NUMERIC.sub("", s)                   # 'abc123def' -> 'abcdef': digits vanish, the letters join
    toks = [w for w in s.split() if w not in STOPWORDS]
    toks = [w for w in toks if len(w) >= min_len]
    return [porter_stem(w) for w in toks] if stem else toks


def preprocess_documents(texts, stem=True):
    """what both load_corpus functions call (reference LabeledLDA.py:45, CascadeLDA.py:48): stemmed tokens."""
    return [preprocess_string(t, stem=stem) for t in texts]


# --------------------------------------------------------------------------------------------
# Porter stemmer (M.F. Porter, "An algorithm for suffix stripping", Program 14(3), 1980)
# --------------------------------------------------------------------------------------------
def _is_cons(w, i):
    c = w[i]
    if c in "aeiou":
        return False
    if c == "y":
        return i == 0 or not _is_cons(w, i - 1)
    return True


def _measure(w):
    """number of VC sequences in w."""
    m, i, n = 0, 0, len(w)
    while i < n and _is_cons(w, i):
        i += 1
    while i < n:
        while i < n and not _is_cons(w, i):
            i += 1
        if i >= n:
            break
        m += 1
        while i < n and _is_cons(w, i):
            i += 1
    return m


def _has_vowel(w):
    return any(not _is_cons(w, i) for i in range(len(w)))


def _double_cons(w):
    return len(w) >= 2 and w[-1] == w[-2] and _is_cons(w, len(w) - 1)


def _cvc(w):
    n = len(w)
    if n < 3:
        return False
    return (_is_cons(w, n - 3) and not _is_cons(w, n - 2) and _is_cons(w, n - 1)
            and w[-1] not in "wxy")


_STEP2 = (("ational", "ate"), ("tional", "tion"), ("enci", "ence"), ("anci", "ance"), ("izer", "ize"),
          ("abli", "able"), ("alli", "al"), ("entli", "ent"), ("eli", "e"), ("ousli", "ous"),
          ("ization", "ize"), ("ation", "ate"), ("ator", "ate"), ("alism", "al"), ("iveness", "ive"),
          ("fulness", "ful"), ("ousness", "ous"), ("aliti", "al"), ("iviti", "ive"), ("biliti", "ble"))
_STEP3 = (("icate", "ic"), ("ative", ""), ("alize", "al"), ("iciti", "ic"), ("ical", "ic"),
          ("ful", ""), ("ness", ""))
_STEP4 = ("al", "ance", "ence", "er", "ic", "able", "ible", "ant", "ement", "ment", "ent", "ion",
          "ou", "ism", "ate", "iti", "ous", "ive", "ize")


def porter_stem(w):
    if len(w) <= 2:
        return w
    # step 1a
    if w.endswith("sses"):
        w = w[:-2]
    elif w.endswith("ies"):
        w = w[:-2]
    elif w.endswith("ss"):
        pass
    elif w.endswith("s"):
        w = w[:-1]
    # step 1b
    flag = False
    if w.endswith("eed"):
        if _measure(w[:-3]) > 0:
            w = w[:-1]
    elif w.endswith("ed") and _has_vowel(w[:-2]):
        w, flag = w[:-2], True
    elif w.endswith("ing") and _has_vowel(w[:-3]):
        w, flag = w[:-3], True
    if flag:
        if w.endswith(("at", "bl", "iz")):
            w += "e"
        elif _double_cons(w) and w[-1] not in "lsz":
            w = w[:-1]
        elif _measure(w) == 1 and _cvc(w):
            w += "e"
    # step 1c
    if w.endswith("y") and _has_vowel(w[:-1]):
        w = w[:-1] + "i"
    # step 2 / 3
    for table in (_STEP2, _STEP3):
        for suf, rep in table:
            if w.endswith(suf):
                if _measure(w[:-len(suf)]) > 0:
                    w = w[:-len(suf)] + rep
                break
    # step 4
    for suf in sorted(_STEP4, key=len, reverse=True):
        if w.endswith(suf):
            stem = w[:-len(suf)]
            if _measure(stem) > 1 and (suf != "ion" or stem.endswith(("s", "t"))):
                w = stem
            break
    # step 5
    if w.endswith("e"):
        stem = w[:-1]
        m = _measure(stem)
        if m > 1 or (m == 1 and not _cvc(stem)):
            w = stem
    if _measure(w) > 1 and _double_cons(w) and w.endswith("l"):
        w = w[:-1]
    return w


# --------------------------------------------------------------------------------------------
# Dictionary
# --------------------------------------------------------------------------------------------
class Dictionary(object):
    """token <-> integer id map with document frequencies (the gensim Dictionary surface the
    reference uses).  Ids are handed out in sorted-token order within each new document, in document
    order; ``filter_extremes`` re-numbers the survivors keeping their relative order."""

    def __init__(self, documents=None):
        self.token2id = {}
        self.id2token = {}
        self.dfs = {}
        self.num_docs = 0
        self.num_pos = 0
        if documents is not None:
            self.add_documents(documents)

    def add_documents(self, documents):
        for doc in documents:
            counter = Counter(doc)
            for tok in sorted(counter):
                if tok not in self.token2id:
                    self.token2id[tok] = len(self.token2id)
                tid = self.token2id[tok]
                self.dfs[tid] = self.dfs.get(tid, 0) + 1
            self.num_docs += 1
            self.num_pos += len(doc)
        self.id2token = {i: t for t, i in self.token2id.items()}

    def __len__(self):
        return len(self.token2id)

    def __getitem__(self, tokenid):
        return self.id2token[tokenid]

    def __iter__(self):
        return iter(self.keys())

    def keys(self):
        return list(self.token2id.values())

    def values(self):
        return [self.id2token[i] for i in self.keys()]

    def items(self):
        return [(i, self.id2token[i]) for i in self.keys()]

    def doc2bow(self, document):
        counter = Counter(map(self.token2id.get, document))     # (ids counted at C speed; unknown tokens map to None)
        counter.pop(None, None)
        return sorted(counter.items())

    def filter_extremes(self, no_below=5, no_above=0.5, keep_n=100000):
        no_above_abs = int(no_above * self.num_docs)
        good = [i for i in self.token2id.values() if no_below <= self.dfs.get(i, 0) <= no_above_abs]
        good.sort(key=lambda i: self.dfs.get(i, 0), reverse=True)
        if keep_n is not None:
            good = good[:keep_n]
        keep = sorted(good)
        remap = {old: new for new, old in enumerate(keep)}
        self.token2id = {t: remap[i] for t, i in self.token2id.items() if i in remap}
        self.dfs = {remap[i]: df for i, df in self.dfs.items() if i in remap}
        self.id2token = {i: t for t, i in self.token2id.items()}
