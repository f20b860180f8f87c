"""The paq8 stage on the MI355X through the C ABI (cmx_p8stage_create / _run): host front end + role kernels, against
columns 434..2024 of the committed traces of the UNMODIFIED reference predictor and against the reference-derived
per-step hashes of tests/golden/make_paq8_hashes.py (32 KB of text, wiki markup, binary records), in ragged chunks.
The same bodies run on the host in tests/test_p8stage_host.py."""
import os

import numpy as np
import pytest

from conftest import load_golden
import make_golden as mg

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(900)]


def run_device(data, chunks, generator_counter=None, pos=None):
    import torch
    from cmix_amd import engine as E
    st = E.P8Stage(0)
    if generator_counter is not None:
        st.set_generator_counter(generator_counter)
    if pos is not None:
        st.debug_set_pos(pos)
    outs, pos, k = [], 0, 0
    data = bytes(data)
    while pos < len(data):
        n = min(chunks[k % len(chunks)], len(data) - pos)
        k += 1
        o = st.run(data[pos:pos + n])
        outs.append(o)
        pos += n
    st.sync()
    got = torch.cat(outs).cpu().numpy()
    st.close()
    return got


@pytest.mark.parametrize("name", ["text_96", "binary_64"])
def test_stage_reproduces_golden_columns(name):
    g = load_golden(name)
    probs = mg.unpack_probs(g)
    got = run_device(g["stream"], [1, 1, 7, 30])
    want = np.ascontiguousarray(probs[:, 434:2025])
    bad = np.argwhere(got.view(np.uint32) != want.view(np.uint32))
    assert bad.size == 0, (name, "first mismatch (step, column):", bad[0], got[tuple(bad[0])] * 4095, want[tuple(bad[0])] * 4095)


@pytest.mark.parametrize("name", ["text_32k", "wiki_12k", "records_8k", "mixed_24k", "rich_16k", "hdrs_4k"])
def test_stage_vs_reference_hashes(name):
    from make_paq8_hashes import row_hash
    from test_p8stage_host import load_hashes
    stream, want = load_hashes(name)
    got = run_device(stream, [1024, 1, 4096, 333])
    h = row_hash(got)
    bad = np.nonzero(h != want)[0]
    assert bad.size == 0, (name, "first differing step:", bad[0], "of", len(want))


@pytest.mark.parametrize("start", [(1 << 31) - 64 * 15000, (1 << 32) - 64 * 15000])
def test_shared_generator_counter_wraps_like_the_references(start):
    """Round 5's 8 MiB finding on the device: the ContextMap family's shared generator has drawn 2^32 values 8.0 MB into enwik-like text, and the kernel's
    look-ahead ring of its values was refilled up to an index compared with `<=` -- wrong in the step the counter wraps, and for good afterwards. The
    counter is placed shortly before 2^31 / 2^32 (a multiple of 64: the same VALUES as a fresh generator, tests/test_p8stage_host.py has the host twin)
    and the stage's 1591 columns must stay the reference's over the wrap."""
    from make_paq8_hashes import row_hash
    from test_p8stage_host import load_hashes
    stream, want = load_hashes("rich_16k")
    got = run_device(stream, [1024, 1, 4096, 333], generator_counter=start)
    bad = np.nonzero(row_hash(got) != want)[0]
    assert bad.size == 0, ("first differing step:", bad[0], "of", len(want))


def test_byte_position_passes_the_end_of_the_history_ring_like_the_references():
    """paq8's byte position passes 2^30, the size of its history ring at level 11, 3000 bytes into this stream (state injection: the fixture is the unmodified
    paq8::Predictor started there, tests/golden/make_paq8_hashes.py pos_1g_6k; tests/test_p8stage_host.py has the host twin): a stream reaches it after 1 GB."""
    from make_paq8_hashes import row_hash
    with np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "paq8_cols_pos_1g_6k.npz")) as z:
        stream, want, pos0 = z["stream"].copy(), z["hash"].copy(), int(z["inject_pos"][0])
    got = run_device(stream, [2990, 10, 1, 1000, 4096], pos=pos0)
    bad = np.nonzero(row_hash(got) != want)[0]
    assert bad.size == 0, ("first differing step:", bad[0], "of", len(want))


@pytest.mark.parametrize("name", ["bmp24_14k", "bmp32_8k", "bmp24_raw_9k", "pgm8_4k", "bmp8_gray_raw_5k", "bmp8_pal_raw_5k",
                                  "wav16s_6k", "wav8s_4k", "wav16m_3k", "wav8m_2k", "pbm1_2k", "bmp1_raw_2k", "bmp4_raw_3k", "jpeg_5k", "jpeg_rst_raw_3k", "mixed_media_12k",
                                  "tga24_5k", "tga_gray_map_32_raw_9k", "jpeg_444_prog_cut_6k", "pam32_thumb_8k", "media_in_text_9k"])
def test_image_model_streams_vs_reference_hashes(name):
    """24 / 32-bit images (im24bitModel): an IMAGE24 block between other blocks, and BMP files inside DEFAULT blocks where paq8's own detector
    switches the model on and off; 8-bit images (im8bitModel): an IMAGE8GRAY block (a PGM), BMP files with a gray and with a colour palette. Chunks that hold image bytes run their roles on one stream, the ContextMap family and the mixer by
    segments (generic kernels / the image model's: cmx_p8s_xfam_kernel, cmx_p8s_xlanes_kernel, cmx_p8s_xmix_kernel), the rnd() stream handed
    over at every switch; the ragged chunk sizes put the switches at chunk starts, ends and in the middle."""
    from make_paq8_hashes import row_hash
    from test_p8stage_host import load_hashes
    stream, want = load_hashes(name)
    got = run_device(stream, [1024, 1, 4096, 333])
    h = row_hash(got)
    bad = np.nonzero(h != want)[0]
    assert bad.size == 0, (name, "first differing step:", bad[0], "of", len(want))


def test_layer0_columns_in_place():
    """Writing straight into columns 434..2024 of a [T, 2078] layer-0 matrix leaves the other columns alone."""
    import torch
    from cmix_amd import engine as E
    g = load_golden("text_96")
    probs = mg.unpack_probs(g)
    st = E.P8Stage(0)
    l0 = torch.full((8 * len(g["stream"]), 2078), -1.0, dtype=torch.float32, device="cuda")
    st.run(bytes(g["stream"]), out=l0, col0=434)
    st.sync()
    got = l0.cpu().numpy()
    st.close()
    assert (got[:, :434] == -1).all() and (got[:, 2025:] == -1).all()
    assert np.array_equal(got[:, 434:2025].view(np.uint32), np.ascontiguousarray(probs[:, 434:2025]).view(np.uint32))


def test_serial_walk_paths_on_device(monkeypatch):
    """CMX_P8CM_SERIAL=1: every ContextMap / ContextMap2 instance is walked serially by one lane each bit (the fallback
    the lane-parallel kernels take on overlaps), then the lanes reload their cached bytes: same values."""
    monkeypatch.setenv("CMX_P8CM_SERIAL", "1")
    g = load_golden("text_96")
    probs = mg.unpack_probs(g)
    got = run_device(g["stream"], [9, 50])
    want = np.ascontiguousarray(probs[:, 434:2025])
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
