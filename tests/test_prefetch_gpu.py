"""GPU: the host->device hand-over of the training loop (xpretrain_amd.utils.prefetch.PrefetchLoader; reference
src/datasets/dataloader.py:95-157): every batch arrives intact, on the device, ordered behind its copy."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("stream", [None, "text"])
def test_prefetch_loader_hands_over_every_batch(stream):
    from xpretrain_amd.utils.prefetch import PrefetchLoader
    g = torch.Generator().manual_seed(3)
    host = [{"video": torch.randn(2, 2, 3, 32, 32, generator=g), "text_input_ids": torch.randint(0, 100, (2, 8), generator=g),
             "meta": "kept as is"} for _ in range(5)]
    loader = PrefetchLoader(host, stream=stream)
    assert len(loader) == 5
    seen = 0
    for want, got in zip(host, loader):
        assert got["video"].is_cuda and got["text_input_ids"].is_cuda and got["meta"] == "kept as is"
        y = got["video"] * 2.0                      # consumed on the current stream right away
        assert torch.equal(y.cpu(), want["video"] * 2.0) and torch.equal(got["text_input_ids"].cpu(), want["text_input_ids"])
        seen += 1
    assert seen == 5
    if stream == "text":
        from xpretrain_amd.modeling.CLIP_ViP import CLIPModel
        assert loader.stream is CLIPModel.shared_text_stream(torch.device("cuda", torch.cuda.current_device()))


def test_prefetch_loader_task_tuples_and_attribute_passthrough():
    """MetaLoader yields (task, batch) (dataloader.py:33-62); attributes of the wrapped loader stay reachable (:153-155)"""
    from xpretrain_amd.utils.prefetch import PrefetchLoader

    class Loader(list):
        n_batches_in_epoch = 7
    src = Loader([("vid", {"x": torch.ones(3)}), ("img", {"x": torch.zeros(3)})])
    out = list(PrefetchLoader(src))
    assert [t for t, _ in out] == ["vid", "img"] and all(b["x"].is_cuda for _, b in out)
    assert PrefetchLoader(src).n_batches_in_epoch == 7


def test_prefetch_on_the_text_stream_copies_behind_the_text_tower():
    """stream='text': the copy of the next batch is armed at hand-over and enqueued when CLIPModel.forward has put the text tower on
    its stream (run_deferred_text_stream_work); without a forward in between, the next hand-over performs it"""
    from xpretrain_amd.modeling.CLIP_ViP import CLIPModel
    from xpretrain_amd.utils.prefetch import PrefetchLoader
    host = [{"x": torch.full((4,), float(i))} for i in range(4)]
    CLIPModel._deferred_text_work.clear()
    it = iter(PrefetchLoader(host, stream="text"))
    b0 = next(it)
    assert len(CLIPModel._deferred_text_work) == 1                  # batch 1 not copied yet
    with torch.cuda.stream(CLIPModel.shared_text_stream(torch.device("cuda", torch.cuda.current_device()))):
        CLIPModel.run_deferred_text_stream_work()                   # what CLIPModel.forward does behind the text tower
    assert not CLIPModel._deferred_text_work
    b1 = next(it)                                                   # (arms batch 2; nobody runs the deferred work this time)
    b2 = next(it)
    b3 = next(it)
    assert len(CLIPModel._deferred_text_work) <= 1                  # superseded requests are withdrawn, not accumulated
    assert [float(b["x"][0]) for b in (b0, b1, b2, b3)] == [0.0, 1.0, 2.0, 3.0]
    with pytest.raises(StopIteration):
        next(it)
    CLIPModel._deferred_text_work.clear()


def test_abandoned_iteration_does_not_leak_its_deferred_copy_into_the_next():
    """with stream='text' an iteration left early (a ``break`` without a following CLIPModel.forward) kept its armed closure --
    bound to the OLD iterator -- in the loader and in CLIPModel._deferred_text_work; the next iteration's first next() then ran it and
    replaced the new epoch's first batch by one of the abandoned iterator.  The closure is withdrawn when an iteration starts or ends."""
    from xpretrain_amd.modeling.CLIP_ViP import CLIPModel
    from xpretrain_amd.utils.prefetch import PrefetchLoader
    host = [{"x": torch.full((4,), float(i))} for i in range(5)]
    CLIPModel._deferred_text_work.clear()
    loader = PrefetchLoader(host, stream="text")
    for b in loader:                       # an eval loop that bails out after one batch
        assert float(b["x"][0]) == 0.0
        break
    assert not CLIPModel._deferred_text_work and loader._armed is None       # (generator closed: finally ran)
    it = iter(loader)
    first = next(it)                       # abandoned WITHOUT closing the generator: the stale closure is still registered ...
    stale = list(CLIPModel._deferred_text_work)
    assert len(stale) == 1
    got = [float(b["x"][0]) for b in loader]                                # ... and a new iteration must not see it
    assert got == [0.0, 1.0, 2.0, 3.0, 4.0]
    stale[0]()                             # a late run of the abandoned closure (a model forward holding it) is a no-op
    assert float(first["x"][0]) == 0.0
    assert [float(b["x"][0]) for b in loader] == [0.0, 1.0, 2.0, 3.0, 4.0]
    CLIPModel._deferred_text_work.clear()
