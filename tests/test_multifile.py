"""The one-process, thread-per-GPU multi-file driver (cmix_amd/multifile.py) on two FAKE engine handles:
assignment, per-GPU ordering, both driver threads really running side by side, error reporting. No GPU."""
import threading

import pytest

from cmix_amd import multifile


class FakeStream:
    log = []
    both_inside = threading.Barrier(2, timeout=20)

    def __init__(self, device, payload, meet=False, fail=False):
        self.device, self.payload, self.fed, self.meet, self.fail = device, payload, 0, meet, fail
        self.closed = False
        FakeStream.log.append(("open", device, len(payload), threading.current_thread().name))

    def feed(self, n):
        if self.meet and self.fed == 0:
            FakeStream.both_inside.wait()   # returns only if the OTHER GPU's thread is inside a feed at the same time
        if self.fail:
            raise ValueError("device fault")
        self.fed += n

    def finish(self):
        assert self.fed >= len(self.payload)
        return b"CMX" + bytes([self.device]) + len(self.payload).to_bytes(4, "big")

    def close(self):
        self.closed = True
        FakeStream.log.append(("close", self.device, len(self.payload), threading.current_thread().name))


def test_two_fake_gpus_longest_first_and_concurrent():
    FakeStream.log = []
    FakeStream.both_inside = threading.Barrier(2, timeout=20)
    files = {"a": b"x" * 5000, "b": b"y" * 3000, "c": b"z" * 2500, "d": b"w" * 100}
    opened = []

    def open_stream(dev, payload):
        first = dev not in [d for d, _ in opened]
        opened.append((dev, len(payload)))
        return FakeStream(dev, payload, meet=first)   # the first file of each GPU waits for the other GPU's thread

    out, rep = multifile.compress_files(files, devices=[0, 1], open_stream=open_stream, step_bytes=1024)
    # longest-first greedy: a -> GPU 0; b -> GPU 1; c -> GPU 1 (3000 < 5000); d -> GPU 0 (5000 < 5500)
    assert rep[0]["files"] == ["a", "d"] and rep[1]["files"] == ["b", "c"]
    assert rep[0]["bytes"] == 5100 and rep[1]["bytes"] == 5500
    assert out["a"] == b"CMX\x00" + (5000).to_bytes(4, "big") and out["c"] == b"CMX\x01" + (2500).to_bytes(4, "big")
    threads = {e[3] for e in FakeStream.log}
    assert threads == {"cmix-gpu0", "cmix-gpu1"}
    assert sum(1 for e in FakeStream.log if e[0] == "close") == 4   # every handle is closed


def test_a_failing_file_is_reported_and_the_other_gpu_finishes():
    FakeStream.log = []

    def open_stream(dev, payload):
        return FakeStream(dev, payload, fail=(len(payload) == 3000))

    files = {"a": b"x" * 5000, "b": b"y" * 3000, "c": b"z" * 2500}
    with pytest.raises(RuntimeError, match="'b' failed on GPU 1"):
        multifile.compress_files(files, devices=[0, 1], open_stream=open_stream, step_bytes=4096)
    closes = [e for e in FakeStream.log if e[0] == "close"]
    assert len(closes) == 3   # a, the failed b, and c after it on the same GPU
