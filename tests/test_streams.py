"""iplan_amd.streams: which of the training cycle's streams share a hardware queue is probed, not left to creation order
(profiles/r06_notes.md section 8)."""
import pytest
import torch


def test_probe_mode_parsing(monkeypatch):
    from iplan_amd import streams
    monkeypatch.delenv("IPLAN_NO_QUEUE_PROBE", raising=False)
    for v, want in (("full", "full"), ("min", "min"), ("verify", "verify"), ("0", "0"), ("nonsense", "0")):
        monkeypatch.setenv("IPLAN_QUEUE_PROBE", v)
        assert streams.probe_mode() == want
    monkeypatch.delenv("IPLAN_QUEUE_PROBE")
    assert streams.probe_mode() == "verify"       # default: creation order, checked after the first cycle


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


@pytest.mark.gpu
def test_verify_replaces_a_side_stream_that_shares_the_main_queue(monkeypatch):
    """harness.SyntheticLoop._verify_queues: after the first cycle the encoder's side streams and the prediction learner's are probed
    against the main stream's queue; a sabotaged assignment (the encoder-BPTT side stream := a stream on the main stream's queue) is
    repaired, a good one is left alone"""
    import contextlib
    import io
    from iplan_amd import ops, streams
    from iplan_amd.config import default_args
    from iplan_amd.harness import SyntheticLoop
    monkeypatch.setenv("IPLAN_QUEUE_PROBE", "verify")
    dev = torch.device("cuda", 0)
    a = default_args("highway", use_cuda=True, max_vehicle_num=9, n_agents=2, episode_limit=20, batch_size_run=4, buffer_size=4, batch_size=3,
                     ppo_epoch=1, pred_batch_size=4)
    loop = SyntheticLoop(a, 4, seed=3, device=dev)
    with contextlib.redirect_stdout(io.StringIO()):
        loop.cycle()
        loop.finish()
    assert loop._queues_verified and isinstance(loop.queue_repairs, list)
    main = torch.cuda.current_stream(dev)
    k_b = (str(dev), main.cuda_stream, 2)
    good = ops._SIDE_STREAMS[k_b]
    assert not streams.shares_queue(main, good, dev)
    # sabotage: find a pool stream that DOES share the main stream's queue and install it as the encoder-BPTT side stream
    bad = None
    for _ in range(24):
        c = torch.cuda.Stream(dev)
        if streams.shares_queue(main, c, dev):
            bad = c
            break
    if bad is None:
        pytest.skip("no pool stream on the main stream's hardware queue on this box")
    ops._SIDE_STREAMS[k_b] = bad
    try:
        assert "enc_bwd" in loop._verify_queues(dev, main)
        assert not streams.shares_queue(main, ops._SIDE_STREAMS[k_b], dev)
        with contextlib.redirect_stdout(io.StringIO()):
            loop.cycle()                                     # ... and the loop runs on the replaced stream
            loop.finish()
        torch.cuda.synchronize()
    finally:
        ops._SIDE_STREAMS[k_b] = good
