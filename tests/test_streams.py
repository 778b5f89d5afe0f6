"""iplan_amd.streams: which of the training cycle's streams share a hardware queue is probed, not left to creation order
(profiles/r06_notes.md section 8)."""
import pytest
import torch


def test_probe_mode_parsing(monkeypatch):
    from iplan_amd import streams
    monkeypatch.delenv("IPLAN_NO_QUEUE_PROBE", raising=False)
    for v, want in (("full", "full"), ("min", "min"), ("0", "0"), ("nonsense", "0")):
        monkeypatch.setenv("IPLAN_QUEUE_PROBE", v)
        assert streams.probe_mode() == want
    monkeypatch.delenv("IPLAN_QUEUE_PROBE")
    assert streams.probe_mode() == "0"            # auto: no process group in this process -> creation order


@pytest.mark.gpu
def test_distinct_stream_does_not_share_the_main_streams_queue(monkeypatch):
    from iplan_amd import streams
    monkeypatch.setenv("IPLAN_QUEUE_PROBE", "full")
    dev = torch.device("cuda", 0)
    main = torch.cuda.current_stream(dev)
    a = torch.cuda.Stream(dev)
    assert streams.shares_queue(a, a, dev), "a stream shares a queue with itself: the probe must see serialisation"
    s1 = streams.distinct_stream(dev, [main])
    assert not streams.shares_queue(main, s1, dev)
    s2 = streams.distinct_stream(dev, [main, s1])
    assert not streams.shares_queue(main, s2, dev) and not streams.shares_queue(s1, s2, dev)
    # with every hardware queue excluded the call still returns a stream (best effort), it does not loop or raise
    many = [main, s1, s2] + [torch.cuda.Stream(dev) for _ in range(8)]
    assert isinstance(streams.distinct_stream(dev, many, tries=3), torch.cuda.Stream)
