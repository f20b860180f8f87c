"""Media at scale on the MI355X, in a file that sorts LAST: both tests were written after round 4's GPU time was spent, so their first run is the
driver's -- behind every other GPU test, where a failure cannot cut the suite short (`pytest -x`). What they check is pinned on the host emulation
(tests/test_p8stage_host.py::test_big_media_stream_digests) and against the reference binary's own file (tests/golden/make_dropin_media.py)."""
import os

import numpy as np
import pytest

from test_gpu_dropin import DROPIN, GOLDEN, _missing, _run

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(900)]


def test_big_media_stream_digests():
    """tests/golden/make_paq8_big_media.py: 168 KB of media as the reference's preprocessor frames them -- a 57 KB IMAGE24 block (the image model's
    segment runs through fourteen 4 KB chunks: the model's family, lanes and mixer kernels launched chunk after chunk with their state in memory in
    between), a 40 KB WAV, a JPEG, a PGM block, 4- and 8-bit BMPs. One digest per 256 steps of the 1591 values against the unmodified reference's."""
    import torch
    from cmix_amd import engine as E
    from make_paq8_hashes import row_hash
    from make_paq8_big_media import digest
    with np.load(os.path.join(GOLDEN, "paq8_big_media_168k.npz")) as z:
        stream, want = bytes(z["stream"]), z["digest"].copy()
    st = E.P8Stage(0)
    hashes = np.zeros(8 * len(stream), np.uint32)
    pos = 0
    while pos < len(stream):
        n = min(4096, len(stream) - pos)
        o = st.run(stream[pos:pos + n])
        st.sync()
        hashes[8 * pos:8 * (pos + n)] = row_hash(o.cpu().numpy())
        pos += n
    st.close()
    bad = np.nonzero(digest(hashes) != want)[0]
    assert bad.size == 0, ("first differing block of 256 steps:", bad[0], "of", len(want))


def test_dropin_engine_file_with_168k_of_media_is_byte_identical():
    """tests/golden/make_dropin_media.py: a 160 x 120 24-bit BMP (the preprocessor's IMAGE24 block: 57 KB, the image model's kernels launched for fourteen
    consecutive chunks), a 40 KB 16-bit stereo WAV, a 320 x 240 JPEG, a 200 x 150 PGM, 4- and 8-bit BMPs between short pieces of text -- every stage of the
    engine on media at scale. The file the unmodified reference binary wrote (10 minutes of its time). Written after round 4's GPU time was spent:
    the paq8 stage's values on this stream are pinned on the host emulation (tests/test_p8stage_host.py::test_big_media_stream_digests)."""
    if not os.path.exists(DROPIN):
        _missing("oracle/_ref/cmix_dropin not built")
    fx = os.path.join(GOLDEN, "dropin_media_168k.npz")
    if not os.path.exists(fx):
        _missing("tests/golden/dropin_media_168k.npz missing (make_dropin_media.py)")
    with np.load(fx) as z:
        payload, blob = z["payload"].tobytes(), z["cmix_file"].tobytes()
    got = _run("-c", [("in", payload)], exe=DROPIN, timeout=900)
    assert len(got) == len(blob) and got == blob, ("sizes", len(got), len(blob))
