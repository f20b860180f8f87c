"""Seeded enwik8-shaped synthetic text (SURVEY.md 8d "S-enwik8[k]").

No corpus and no dictionary file is available on the GPU box, so the word list
itself is synthesised: `n_words` pseudo-English words built from a seeded
syllable grammar, sampled Zipf(s=1).  On top of the word stream the generator
lays the features that shape enwik8's byte statistics for cmix: sentence
structure, `[[wiki links]]`, 4-digit numbers, `''`/`==` markup lines, periodic
`<page>...</page>` XML frames, 2 % UTF-8 two-byte sequences and `&quot;` /
`&amp;` entities, so that the number of distinct byte values V is ~205 (the
LSTM cost depends on V, reference src/predictor.cpp:185-191).

Deterministic for a given (seed, nbytes): shard k of config 5 uses seed 1000+k.
"""
import numpy as np

_ONSETS = ["", "b", "c", "d", "f", "g", "h", "j", "k", "l", "m", "n", "p", "r", "s", "t",
           "v", "w", "y", "z", "th", "sh", "ch", "st", "tr", "pr", "br", "gr", "cl", "pl",
           "qu", "sp", "fl", "sl", "cr", "dr", "wh", "sc", "sm", "sn"]
_NUCLEI = ["a", "e", "i", "o", "u", "ea", "ou", "ai", "io", "ee", "oo", "ie", "au", "y"]
_CODAS = ["", "", "", "n", "r", "s", "t", "l", "d", "m", "ng", "nt", "st", "ck", "x", "ll",
          "ss", "nd", "rt", "ve", "ce", "ly", "ty", "on", "er", "ed", "es", "al"]


def _wordlist(rng, n_words):
    words, seen = [], set()
    while len(words) < n_words:
        nsyl = 1 + int(rng.integers(0, 4) * rng.random() * 1.2)
        w = "".join(_ONSETS[rng.integers(len(_ONSETS))] + _NUCLEI[rng.integers(len(_NUCLEI))] +
                    _CODAS[rng.integers(len(_CODAS))] for _ in range(max(1, nsyl)))
        if w not in seen:
            seen.add(w)
            words.append(w)
    # frequent words are short, as in natural language
    words.sort(key=len)
    head = words[:2000]
    rng.shuffle(head)
    return head + words[2000:]


def enwik_like(nbytes: int, seed: int = 1000, n_words: int = 44515) -> bytes:
    rng = np.random.default_rng(seed)
    words = _wordlist(np.random.default_rng(7), n_words)     # same lexicon for every shard
    ranks = np.arange(1, n_words + 1, dtype=np.float64)
    cdf = np.cumsum(1.0 / ranks)
    cdf /= cdf[-1]
    out, size, page_id, since_page = [], 0, 1, 1 << 30
    terms = [". ", ". ", ".\n", "? ", ", ", ".\n\n"]
    # 100 two-byte UTF-8 code points (Latin-1 supplement / Latin Extended-A / Greek / Cyrillic leads)
    utf = [chr(c) for c in list(range(0xC0, 0x100)) + list(range(0x391, 0x3A1)) +
           list(range(0x410, 0x424))]
    while size < nbytes:
        if since_page > 4096:
            title = " ".join(words[int(np.searchsorted(cdf, rng.random()))].capitalize()
                             for _ in range(int(rng.integers(1, 4))))
            s = ("  <page>\n    <title>%s</title>\n    <id>%d</id>\n    <revision>\n"
                 "      <text xml:space=\"preserve\">" % (title, page_id))
            if page_id > 1:
                s = "</text>\n    </revision>\n  </page>\n" + s
            page_id += 1
            since_page = 0
        else:
            n = int(rng.integers(5, 26))
            idx = np.searchsorted(cdf, rng.random(n))
            ws = [words[int(i)] for i in idx]
            ws[0] = ws[0].capitalize()
            r = rng.random(6)
            if r[0] < 0.15:
                k = int(rng.integers(n))
                ws[k] = "[[" + ws[k] + "]]"
            if r[1] < 0.10:
                ws.append(str(int(rng.integers(1000, 10000))))
            if r[2] < 0.02 * n:
                k = int(rng.integers(n))
                ws[k] = ws[k] + utf[int(rng.integers(len(utf)))] + utf[int(rng.integers(len(utf)))]
            if r[3] < 0.05:
                k = int(rng.integers(n))
                ws[k] = "&quot;" + ws[k] + "&quot;"
            if r[4] < 0.03:
                ws.insert(int(rng.integers(1, n)), "&amp;")
            s = " ".join(ws) + terms[int(rng.integers(len(terms)))]
            if r[5] < 0.04:
                s = "\n== " + ws[0] + " ==\n" + s
            elif r[5] < 0.08:
                s = "''" + s.rstrip() + "''\n"
            elif r[5] < 0.10:
                s = "* " + s.rstrip() + "\n"
        b = s.encode("utf-8")
        out.append(b)
        size += len(b)
        since_page += len(b)
    return b"".join(out)[:nbytes]


def book_like(nbytes: int = 768771, seed: int = 12345) -> bytes:
    """S-book1: plain prose without markup (config 1 sized input)."""
    rng = np.random.default_rng(seed)
    words = _wordlist(np.random.default_rng(7), 44515)
    ranks = np.arange(1, len(words) + 1, dtype=np.float64)
    cdf = np.cumsum(1.0 / ranks)
    cdf /= cdf[-1]
    terms = [". ", ". ", ".\n", "? ", ", ", ".\n\n"]
    out, size = [], 0
    while size < nbytes:
        n = int(rng.integers(5, 26))
        ws = [words[int(i)] for i in np.searchsorted(cdf, rng.random(n))]
        ws[0] = ws[0].capitalize()
        b = (" ".join(ws) + terms[int(rng.integers(len(terms)))]).encode()
        out.append(b)
        size += len(b)
    return b"".join(out)[:nbytes]
