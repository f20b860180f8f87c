"""Seeded enwik8-shaped synthetic text (SURVEY.md 8d "S-enwik8[k]").

No corpus and no dictionary file is available on the GPU box, so the word list
itself is synthesised: `n_words` pseudo-English words built from a seeded
syllable grammar, sampled Zipf(s=1).  On top of the word stream the generator
lays the features that shape enwik8's byte statistics for cmix: sentence
structure, `[[wiki links]]`, 4-digit numbers, `''`/`==` markup lines, periodic
`<page>...</page>` XML frames, 2 % UTF-8 two-byte sequences and `&quot;` /
`&amp;` entities. With `rich=True` (the bench shard) the alphabet is enwik8's: every
printable ASCII character through wiki / URL / table markup tokens and two-, three- and
four-byte UTF-8 from 46 script blocks (interlanguage links, foreign words), so that the
number of distinct byte values is V = 205 (the LSTM cost depends on V, reference
src/predictor.cpp:185-191); without it V is ~145 (the small parity fixtures keep that form).

Deterministic for a given (seed, nbytes): shard k of config 5 uses seed 1000+k.
"""
import re

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


# (first code point, count) of the script blocks the rich alphabet draws from: UTF-8 lead bytes C2..DE except DD,
# E0..ED, EF, F0 -- with the 64 continuation bytes and 97 ASCII values that is V = 205, enwik8's vocabulary size
_BLOCKS = [(0xA1, 31), (0xC0, 64), (0x100, 64), (0x140, 64), (0x180, 64), (0x1C0, 64), (0x200, 64), (0x250, 48), (0x280, 48),
           (0x2C0, 32), (0x300, 64), (0x340, 48), (0x391, 47), (0x3C0, 10), (0x410, 48), (0x440, 16), (0x490, 48),
           (0x4C0, 32), (0x531, 15), (0x561, 31), (0x580, 7), (0x5D0, 27), (0x621, 26), (0x641, 10), (0x6A0, 32),
           (0x6C0, 20), (0x710, 32), (0x780, 38), (0x905, 53), (0xE01, 58), (0x10D0, 33), (0x2013, 20), (0x3041, 83),
           (0x30A1, 90), (0x4E00, 4096), (0x5E00, 4096), (0x6E00, 4096), (0x7E00, 4096), (0x8E00, 4096), (0x9000, 4096),
           (0xAC00, 4096), (0xBC00, 4096), (0xC800, 2048), (0xD000, 1024), (0xFF01, 94), (0x10330, 27)]
_LANGS = ["de", "fr", "ja", "ru", "zh", "ar", "he", "el", "ko", "pl", "th", "ka", "hy", "hi", "sv", "es", "nl", "eo"]
_MARKUP = ["{{cite web|url=http://www.%s.org/~%s/index.php?id=%d&amp;p=%d#ref_%d|title=%s}}", "<ref name=\"%s\">%s, p. %d</ref>",
           "{| class=\"wikitable\"\n! %s !! %s\n|-\n| %d || %d\n|}", "$%d.%d", "%d%%", "%s@%s.com", "C:\\%s\\%s", "`%s`",
           "%s_%s", "[%d]", "%d+%d=%d", "%s^%d", "(%s; %s!)", "\t%s", "#REDIRECT [[%s]]", "<!-- %s -->", "\'\'\'%s\'\'\'", "~%d",
           "%s/%s", "%d:%d", "[http://%s.com %s]", "&lt;%s&gt;", "%d*%d", "{%s}", "X%d", "Q%d", "Z%s"]


def _rich_token(rng, word):
    f = _MARKUP[int(rng.integers(len(_MARKUP)))]
    args = tuple(word() if c == "s" else int(rng.integers(1, 2000)) for c in re.findall(r"%([sd])", f.replace("%%", "")))
    return f % args


def _foreign(rng):
    a, n = _BLOCKS[int(rng.integers(len(_BLOCKS)))]
    return "".join(chr(a + int(rng.integers(n))) for _ in range(int(rng.integers(2, 7))))


def enwik_like(nbytes: int, seed: int = 1000, n_words: int = 44515, rich: bool = False, lexicon=None) -> bytes:
    """`lexicon`: a list of words to draw from instead of the synthesised one (config 3's fixture uses the reference's
    english.dic so that the WRT transform finds its words; the payload is then stored in the fixture)."""
    rng = np.random.default_rng(seed)
    if lexicon is not None:
        words, n_words = list(lexicon), len(lexicon)
    else:
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
            if rich and page_id > 1:
                s = "".join("[[%s:%s]]\n" % (_LANGS[int(rng.integers(len(_LANGS)))], _foreign(rng))
                            for _ in range(int(rng.integers(0, 5)))) + s
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
            if rich:
                rr = rng.random(2)
                if rr[0] < 0.12:
                    ws.insert(int(rng.integers(1, len(ws) + 1)),
                              _rich_token(rng, lambda: words[int(np.searchsorted(cdf, rng.random()))]))
                if rr[1] < 0.10:
                    ws.insert(int(rng.integers(1, len(ws) + 1)), _foreign(rng))
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


# ---- S-silesia: twelve members in the flavours of the Silesia corpus ----------------------
SILESIA_NOMINAL = {"dickens": 10192446, "mozilla": 51220480, "mr": 9970564, "nci": 33553445, "ooffice": 6152192,
                   "osdb": 10085684, "reymont": 6627202, "samba": 21606400, "sao": 7251944, "webster": 41458703,
                   "xml": 5345280, "x-ray": 8474240}


def _records(nbytes, seed, reclen=28):
    """Little-endian record table: id, two slowly varying 32-bit fields, flags, a 12-byte low-entropy tail."""
    r = np.random.default_rng(seed)
    n = nbytes // reclen + 1
    rec = np.zeros((n, reclen), np.uint8)
    ids = np.arange(n, dtype=np.uint32) + 1000
    a = np.cumsum(r.integers(-3, 40, n)).astype(np.uint32)
    b = (r.integers(0, 1 << 14, n) * 4).astype(np.uint32)
    for k in range(4):
        rec[:, k] = (ids >> (8 * k)) & 255
        rec[:, 4 + k] = (a >> (8 * k)) & 255
        rec[:, 8 + k] = (b >> (8 * k)) & 255
    rec[:, 12] = r.integers(0, 4, n)
    rec[:, 16:reclen] = r.integers(32, 48, (n, reclen - 16))
    return rec.tobytes()[:nbytes]


def _opcode_soup(nbytes, seed):
    """x86-like code: short register instructions and relative calls / jumps to a few targets (the reference's
    detector makes EXE blocks of it and its e8e9 transform rewrites the addresses)."""
    r = np.random.default_rng(seed)
    ops = [0x89, 0x8B, 0x01, 0x03, 0x50, 0x58, 0x74, 0x75, 0x83, 0xC7, 0x55, 0x5D, 0xC3, 0x90, 0x31, 0x85]
    targets = [int(t) for t in r.integers(0x100, max(0x200, nbytes), 24)]
    code = bytearray()
    while len(code) < nbytes:
        for _ in range(int(r.integers(2, 7))):
            code += bytes([ops[int(r.integers(len(ops)))], int(r.integers(0, 0xE0))])
        t = targets[int(r.integers(len(targets)))]
        code += bytes([0xE8 if r.random() < 0.8 else 0xE9]) + ((t - (len(code) + 5)) & 0xFFFFFFFF).to_bytes(4, "little")
    return bytes(code[:nbytes])


def _xml_like(nbytes, seed):
    r = np.random.default_rng(seed)
    w = _wordlist(np.random.default_rng(7), 3000)
    out, size, i = [b"<?xml version=\"1.0\" encoding=\"UTF-8\"?>\n<catalog>\n"], 50, 0
    while size < nbytes:
        i += 1
        s = ("  <item id=\"%d\" type=\"%s\">\n    <name>%s %s</name>\n    <price currency=\"EUR\">%d.%02d</price>\n"
             "    <tags>%s</tags>\n  </item>\n" % (i, w[int(r.integers(40))], w[int(r.integers(3000))].capitalize(),
                                                 w[int(r.integers(3000))], int(r.integers(1, 900)), int(r.integers(100)),
                                                 ",".join(w[int(k)] for k in r.integers(0, 300, int(r.integers(1, 5)))))).encode()
        out.append(s)
        size += len(s)
    return b"".join(out)[:nbytes]


def silesia_like(scale_bytes: int = 0, seed: int = 4000):
    """{name: payload}: twelve members named and proportioned like the Silesia corpus. scale_bytes = 0 gives the nominal
    sizes; otherwise every member is scaled so that the largest has `scale_bytes` bytes (at least 2 KB each). Flavours:
    prose (dickens, reymont, webster), wiki-like text with markup (nci), XML (xml), mixed text + records (samba, osdb),
    x86-like code with data sections (mozilla, ooffice), smooth 16-bit samples (mr, x-ray), uniform-ish bytes (sao)."""
    big = max(SILESIA_NOMINAL.values())
    out = {}
    for k, (name, nominal) in enumerate(sorted(SILESIA_NOMINAL.items())):
        n = nominal if not scale_bytes else max(2048, nominal * scale_bytes // big)
        s = seed + k
        if name in ("dickens", "reymont", "webster"):
            d = book_like(n, s)
        elif name == "nci":
            d = enwik_like(n, s)
        elif name == "xml":
            d = _xml_like(n, s)
        elif name in ("samba", "osdb"):
            h = n // 2
            d = enwik_like(h, s) + _records(n - h, s)
        elif name in ("mozilla", "ooffice"):
            a = n * 2 // 3
            d = _opcode_soup(a, s) + _records(n - a, s, 20)
        elif name in ("mr", "x-ray"):
            r = np.random.default_rng(s)
            v = (2048 + np.cumsum(r.integers(-6, 7, n // 2 + 1))).astype(np.int64) & 0x0FFF
            d = v.astype("<u2").tobytes()[:n]
        else:  # sao: star catalogue -> close to uniform bytes with a little record structure
            r = np.random.default_rng(s)
            d = bytes(r.integers(0, 256, n, dtype=np.uint8))
        out[name] = d
    return out
