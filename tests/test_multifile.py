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


def test_compress_paths_runs_one_process_per_file_bound_to_its_gpu(tmp_path):
    """The process-per-file driver with a FAKE engine command line (a shell script that records CMIX_DEVICE and its arguments):
    longest-first assignment, the device binding through the environment, the optional dictionary argument, failure reporting."""
    import os
    import stat
    exe = tmp_path / "fake_engine.sh"
    exe.write_text('#!/bin/sh\n'
                   'eval "out=\\${$#}"\n'
                   'echo "$CMIX_DEVICE $*" > "$out"\n'
                   'case "$*" in *poison*) exit 3;; esac\n')
    exe.chmod(exe.stat().st_mode | stat.S_IEXEC)
    jobs = []
    for name, n in (("a", 5000), ("b", 3000), ("c", 2500), ("d", 100)):
        src = tmp_path / name
        src.write_bytes(b"x" * n)
        jobs.append((str(src), str(tmp_path / (name + ".cmix"))))
    rep = multifile.compress_paths(jobs, devices=[0, 1], exe=str(exe), dictionary=str(tmp_path / "dic"))
    assert [os.path.basename(f) for f in rep[0]["files"]] == ["a", "d"] and [os.path.basename(f) for f in rep[1]["files"]] == ["b", "c"]
    got = {os.path.basename(dst)[0]: open(dst).read().split() for _, dst in jobs}
    assert got["a"][0] == "0" and got["d"][0] == "0" and got["b"][0] == "1" and got["c"][0] == "1"
    assert got["b"][1:] == ["-c", str(tmp_path / "dic"), jobs[1][0], jobs[1][1]]
    bad = tmp_path / "poison"
    bad.write_bytes(b"y" * 4000)
    with pytest.raises(RuntimeError, match="poison.*failed on GPU 1"):
        multifile.compress_paths(jobs + [(str(bad), str(tmp_path / "poison.cmix"))], devices=[0, 1], exe=str(exe))
    with pytest.raises(FileNotFoundError):
        multifile.compress_paths(jobs, devices=[0], exe=str(tmp_path / "missing"))
    with pytest.raises(ValueError, match="open_stream is required"):
        multifile.compress_files({"a": b"x"}, devices=[0], open_stream=None)
