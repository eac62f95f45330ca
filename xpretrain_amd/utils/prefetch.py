"""``PrefetchLoader``: the host->device hand-over of the reference's training loop (``src/datasets/dataloader.py:79-157``: batch
i+1 is copied to the GPU on a side stream while step i computes; ``next()`` makes the compute stream wait for the copy and
``record_stream``s the tensors).  Same class name, constructor and iteration protocol, so ``run_pretrain.py:245-247`` keeps its lines.

Differences, all on the copy side: host tensors are pinned once per batch (a pageable source makes ``cuda(non_blocking=True)`` a
synchronous copy), and the copy stream can be handed in (``stream=``): ``stream="text"`` reuses the stream ``CLIPModel.forward`` runs
the text tower on -- the step then touches no more HIP streams than it does without a loader (beyond four streams
the runtime maps streams onto shared hardware queues and the step's overlap degrades); the copy of batch i+1 is then enqueued right
behind the text tower's forward of step i (``CLIPModel.defer_to_text_stream``), where that stream idles until the backward.
``stream="auto"``: the text tower's stream when a process group with more than one rank exists (its collectives already own a
stream), a stream of the loader's own otherwise.  Measured: profiles/r05g_prefetch_stream_environment.txt."""
import torch


def _map(batch, f):
    if isinstance(batch, torch.Tensor):
        return f(batch)
    if isinstance(batch, list):
        return [_map(t, f) for t in batch]
    if isinstance(batch, tuple):
        return tuple(_map(t, f) for t in batch)
    if isinstance(batch, dict):
        return {k: _map(v, f) for k, v in batch.items()}
    return batch


def move_to_cuda(batch):
    """dataloader.py:66-77 (pinning the source first, so the copy really is asynchronous)"""
    return _map(batch, lambda t: (t if t.is_cuda or t.is_pinned() else t.pin_memory()).cuda(non_blocking=True))


def record_cuda_stream(batch):
    """dataloader.py:80-90"""
    cur = torch.cuda.current_stream()
    _map(batch, lambda t: (t.record_stream(cur), t)[1] if t.is_cuda else t)


class PrefetchLoader(object):
    def __init__(self, loader, img_normalize=None, stream=None):
        self.loader = loader
        self.img_normalize = img_normalize
        self._stream_arg = stream
        self.stream = None
        self._defer = None          # set with the text tower's stream: enqueue the next copy behind the tower's forward
        self._armed = None
        self._cancel = None

    def _copy_stream(self):
        if self.stream is None:
            s = self._stream_arg
            if isinstance(s, torch.cuda.Stream):
                self.stream = s
            elif s == "text" or (s == "auto" and self._data_parallel()):
                from ..modeling.CLIP_ViP import CLIPModel
                self.stream = CLIPModel.shared_text_stream(torch.device("cuda", torch.cuda.current_device()))
                self._defer, self._cancel = CLIPModel.defer_to_text_stream, CLIPModel.cancel_deferred_text_stream_work
            else:
                self.stream = torch.cuda.Stream()
        return self.stream

    def _disarm(self):
        """Drop a deferred copy nobody ran (an iteration abandoned without a following CLIPModel.forward: a ``break`` or an exception in
        a loop that only calls forward_video / forward_text).  Left armed, its closure -- bound to the OLD iterator -- would run inside
        the next iteration's first ``next()`` and replace that iteration's first batch."""
        armed, self._armed = self._armed, None
        if armed is not None and self._cancel is not None:
            self._cancel(armed)

    def __iter__(self):
        self._disarm()
        loader_it = iter(self.loader)
        try:
            self.preload(loader_it)
            batch = self.next(loader_it)
            while batch is not None:
                yield batch
                batch = self.next(loader_it)
        finally:
            self._disarm()

    def __len__(self):
        return len(self.loader)

    def preload(self, it):
        try:
            self.batch = next(it)
        except StopIteration:
            self.batch = None
            return
        with torch.cuda.stream(self._copy_stream()):
            if isinstance(self.batch, tuple) and len(self.batch) == 2 and isinstance(self.batch[0], str):     # (task, batch) of MetaLoader
                self.batch = (self.batch[0], move_to_cuda(self.batch[1]))
            else:
                self.batch = move_to_cuda(self.batch)

    @staticmethod
    def _data_parallel():
        import torch.distributed as dist
        from .. import distributed as D
        return (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1) or D.FORCE_COLLECTIVES

    def next(self, it):
        if self._armed is not None:          # no model forward ran since the last hand-over: copy now
            armed = self._armed
            self._cancel(armed)
            armed()
        torch.cuda.current_stream().wait_stream(self._copy_stream())
        batch = self.batch
        if batch is not None:
            record_cuda_stream(batch)
        if self._defer is not None and batch is not None:
            def once():
                if self._armed is once:          # (a closure of an abandoned iteration is no longer the armed one: it does nothing)
                    self._armed = None
                    self.preload(it)
            self._armed = once
            self._defer(once)
        else:
            self.preload(it)
        return batch

    def __getattr__(self, name):
        if name in ("loader", "stream", "batch"):          # (not set yet: e.g. while unpickling -- do not recurse)
            raise AttributeError(name)
        return self.loader.__getattribute__(name)
