"""CPU: the paq8 stage's HOST front end (cmix_amd/csrc/p8front/p8f_{stem,text,word,xml,record,match,exe,ctxmodels,lpm}.c) is a prefix-renamed twin of the
oracle's restatement (oracle/paq8_*.c): nothing is linked across the line, so the two can drift, and a comparison of one with the other proves
nothing about either. Two guards (round-4 review, "What's weak" 12):
  * the twins stay twins: comment- and prefix-normalised, the line difference of each pair may not grow past what it is today;
  * the PRODUCT objects themselves are pinned where they can run alone: the stemmer tests of tests/test_oracle_paq8core.py, which compare the oracle with the
    unmodified reference's classes (oracle/_ref/libcmixrefpaq8.so), run again with libcmixamd.so's p8f_* entry points in the oracle's place."""
import difflib
import os
import re

import pytest

from conftest import ROOT
from oracle import oracle as O
from oracle import refharness as R

# pair -> the normalised line difference on the day this test was written (round 5); a change to one file that is not mirrored in the other raises it
PAIRS = {"stem": 2, "text": 2, "word": 1, "xml": 0, "record": 0, "match": 9, "exe": 1, "ctxmodels": 30, "lpm": 15}


def _norm(path, product):
    s = open(path).read()
    s = re.sub(r"/\*.*?\*/", "", s, flags=re.S)
    s = re.sub(r"//[^\n]*", "", s)
    for a in (("p8f_", "P8F_") if product else ("orc_p8_", "ORC_P8_", "orc_", "ORC_")):
        s = s.replace(a, "X_")
    return [x for x in (re.sub(r"\s+", " ", l).strip() for l in s.split("\n")) if x]


@pytest.mark.parametrize("name", sorted(PAIRS))
def test_front_end_and_oracle_twin_have_not_drifted(name):
    a = _norm(os.path.join(ROOT, "cmix_amd", "csrc", "p8front", "p8f_%s.c" % name), True)
    b = _norm(os.path.join(ROOT, "oracle", "paq8_%s.c" % name), False)
    same = sum(m.size for m in difflib.SequenceMatcher(None, a, b, autojunk=False).get_matching_blocks())
    diff = max(len(a), len(b)) - same
    assert diff <= PAIRS[name], "p8f_%s.c and oracle/paq8_%s.c differ in %d normalised lines (%d when the guard was written): mirror the change in the twin" % (name, name, diff, PAIRS[name])


class _ProductAsOracle:
    """the product library answering to the oracle's names: orc_p8_x -> p8f_x (same signatures: the files are twins)"""

    def __init__(self, product, oracle):
        self._p, self._o = product, oracle

    def __getattr__(self, name):
        if name.startswith("orc_p8_") and hasattr(self._p, "p8f_" + name[7:]):
            return getattr(self._p, "p8f_" + name[7:])
        return getattr(self._o, name)


needs_ref = pytest.mark.skipif(not R.paq8core_available(), reason="oracle/_ref/libcmixrefpaq8.so not built")


@needs_ref
@pytest.mark.parametrize("case", ["test_english_stemmer_vs_reference", "test_french_and_german_stemmers_vs_reference"])
def test_product_front_end_objects_vs_reference(case, monkeypatch):
    """The stemmers are the self-contained part of the front end (stem letters, flags, both hash sets of every word against the reference's EnglishStemmer /
    FrenchStemmer / GermanStemmer). The sub-models (word, text, XML, record, exe, match, linear prediction) hand their contexts to the emitter of the chunk
    being built instead of to ContextMaps of their own, so they cannot run stand-alone; they are pinned one level up, by the per-step hashes of the unmodified
    paq8::Predictor through the stage's host emulation (tests/test_p8stage_host.py::test_stage_vs_reference_hashes) and on the device."""
    import ctypes as C
    import test_oracle_paq8core as T
    from cmix_amd import build
    product = C.CDLL(build.build())
    proxy = _ProductAsOracle(product, O.lib())
    used = []
    orig = _ProductAsOracle.__getattr__

    def spy(self, name):
        if name.startswith("orc_p8_") and hasattr(self._p, "p8f_" + name[7:]):
            used.append(name)
        return orig(self, name)
    monkeypatch.setattr(_ProductAsOracle, "__getattr__", spy)
    monkeypatch.setattr(T.O, "lib", lambda: proxy)
    with O.scope():
        getattr(T, case)()
    assert used, "%s never called a product entry point" % case
