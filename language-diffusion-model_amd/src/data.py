"""Device-side batch production for the training step's batch contract.

The step consumes ``{"input_ids": int64 [B, S], "attention_mask": int64 [B, S]}`` (reference:
src/dataset.py:48,126-127 feeds src/trainer.py:120-121).  Three producers, all ending in device
tensors of exactly that form:

* ``collate_batch_device`` -- ``collate_batch`` of the reference (src/dataset.py:26-48) with the
  padding done on the device: the unpadded tokens go to pinned memory and over the host link once,
  a kernel writes the padded ids and the mask;
* ``DevicePrefetcher`` -- wraps any iterable of host batches (e.g. the reference's DataLoader) and
  keeps the next batches' copies in flight on a side stream while the current step runs;
* ``SyntheticTokenBatches`` -- batches generated on the device (Philox), optionally with ragged
  lengths and pad tails, for benchmarks without an input pipeline.
"""

from __future__ import annotations

from collections import deque

import torch

from . import _native as N


def collate_batch_device(batch, device=None, pad_id: int = 0):
    """``collate_batch`` (src/dataset.py:26-48) producing device tensors.

    ``batch``: list of items with 1-D int64 ``input_ids`` (their ``attention_mask`` is all ones in
    the reference, src/tokenizer.py -> src/dataset.py:126-127, so only the lengths are used)."""
    device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    lens = [int(item["input_ids"].numel()) for item in batch]
    B, smax, total = len(batch), (max(lens) if lens else 0), sum(lens)
    ids = torch.empty(B, smax, dtype=torch.int64, device=device)
    mask = torch.empty(B, smax, dtype=torch.int64, device=device)
    if B == 0 or smax == 0:
        return {"input_ids": ids, "attention_mask": mask}
    staging = torch.empty(total + B + 1, dtype=torch.int64, pin_memory=True)
    off = 0
    for i, item in enumerate(batch):
        staging[B + 1 + off:B + 1 + off + lens[i]] = item["input_ids"].reshape(-1)
        staging[i] = off
        off += lens[i]
    staging[B] = off
    dev_buf = staging.to(device, non_blocking=True)
    with torch.cuda.device(device):
        N.check(N.lib().ldm_collate_pad(dev_buf[B + 1:].data_ptr(), dev_buf.data_ptr(), B, smax, pad_id,
                                        ids.data_ptr(), mask.data_ptr(), N.stream()), "ldm_collate_pad")
    return {"input_ids": ids, "attention_mask": mask}


class DevicePrefetcher:
    """Iterates ``loader`` (host batches: dicts of CPU tensors, or lists of un-collated items) and
    yields device batches, with up to ``depth`` batches' host-to-device copies queued on a side
    stream ahead of the consumer."""

    def __init__(self, loader, device=None, depth: int = 2, pad_id: int = 0):
        self.loader = loader
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.depth = max(1, depth)
        self.pad_id = pad_id
        self.stream = torch.cuda.Stream(device=self.device)

    def __len__(self):
        return len(self.loader)

    def _stage(self, host_batch):
        with torch.cuda.stream(self.stream):
            if isinstance(host_batch, dict):
                out = {}
                for k, v in host_batch.items():
                    if torch.is_tensor(v):
                        src = v if v.is_pinned() else v.pin_memory()
                        out[k] = src.to(self.device, non_blocking=True)
                    else:
                        out[k] = v
            else:
                out = collate_batch_device(host_batch, self.device, self.pad_id)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        return out, ev

    def __iter__(self):
        it = iter(self.loader)
        queue: deque = deque()
        done = False
        while True:
            while not done and len(queue) < self.depth:
                try:
                    queue.append(self._stage(next(it)))
                except StopIteration:
                    done = True
            if not queue:
                return
            batch, ev = queue.popleft()
            torch.cuda.current_stream(self.device).wait_event(ev)
            for v in batch.values():
                if torch.is_tensor(v):
                    v.record_stream(torch.cuda.current_stream(self.device))
            yield batch


class SyntheticTokenBatches:
    """``num_batches`` synthetic batches made on the device: ids uniform in [id_lo, vocab_size), BOS in
    column 0, EOS at the end of each sequence, and -- with ``min_len < seq_len`` -- ragged lengths
    with pad tails and a matching attention mask.  Batch ``i`` is a pure function of (seed, i)."""

    def __init__(self, num_batches: int, batch_size: int, seq_len: int, vocab_size: int, seed: int = 1234,
                 device=None, id_lo: int = 1000, bos_id: int = 101, eos_id: int = 102, pad_id: int = 0,
                 min_len: int | None = None):
        self.num_batches, self.batch_size, self.seq_len, self.vocab_size = num_batches, batch_size, seq_len, vocab_size
        self.seed, self.id_lo = seed, min(id_lo, vocab_size - 1)
        self.bos_id, self.eos_id, self.pad_id = bos_id, eos_id, pad_id
        self.min_len = seq_len if min_len is None else min_len
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())

    def __len__(self):
        return self.num_batches

    def batch(self, index: int):
        ids = torch.empty(self.batch_size, self.seq_len, dtype=torch.int64, device=self.device)
        mask = torch.empty_like(ids)
        with torch.cuda.device(self.device):
            N.check(N.lib().ldm_synth_tokens(self.batch_size, self.seq_len, self.id_lo, self.vocab_size, self.bos_id,
                                             self.eos_id, self.pad_id, self.min_len, self.seed, index, ids.data_ptr(),
                                             mask.data_ptr(), N.stream()), "ldm_synth_tokens")
        return {"input_ids": ids, "attention_mask": mask}

    def __iter__(self):
        for i in range(self.num_batches):
            yield self.batch(i)
