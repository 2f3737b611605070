"""Question dataset and loaders feeding the training step's batch contract.

Mirror of the reference's ``src/dataset.py`` (``collate_batch`` :26-48, ``QuestionDataset`` :51-134,
``create_question_dataloaders`` :137-202): same names, arguments, split and batch contents.  The
tokenizer is whatever object the caller passes, used through
``tokenizer.encode(text, max_length=..., return_tensors=False) -> {"input_ids": [...], "attention_mask": [...]}``
(src/dataset.py:122-124).

One addition: ``create_question_dataloaders(..., device=...)`` returns loaders that pad on the GPU --
the workers hand over the un-collated items, ``DevicePrefetcher`` ships the unpadded tokens once and
``ldm_collate_pad`` writes the padded ``input_ids`` / ``attention_mask`` in HBM, with the next
batches' copies in flight while the current step runs.  Without ``device`` the loaders yield the
reference's host batches.
"""

from __future__ import annotations

from collections.abc import Callable

import torch
from torch.utils.data import DataLoader, Dataset, random_split

from .data import DevicePrefetcher

FIELDS = ("input_ids", "attention_mask")


def collate_batch(batch):
    """Right-pads every field with 0 to the longest item of the batch (src/dataset.py:26-48)."""
    longest = max((int(item[FIELDS[0]].numel()) for item in batch), default=0)
    out = {}
    for name in FIELDS:
        first = batch[0][name] if batch else torch.zeros(0, dtype=torch.long)
        padded = first.new_zeros((len(batch), longest))
        for row, item in zip(padded, batch):
            row[:item[name].numel()] = item[name]
        out[name] = padded
    return out


def _items(batch):
    """collate_fn of the device-padding loaders: the items travel un-collated."""
    return batch


class QuestionDataset(Dataset):
    """One question per non-empty line of ``file_path``: the text between the first and the second
    ``@@`` of the line; lines without an ``@@`` are skipped (src/dataset.py:83-101).  Items are
    un-padded 1-D int64 ``input_ids`` / ``attention_mask`` (src/dataset.py:107-134)."""

    def __init__(self, file_path: str, tokenizer, max_length: int = 512,
                 transform: Callable[[dict], dict] | None = None):
        self.file_path = file_path
        self.tokenizer = tokenizer
        self.max_length = max_length
        self.transform = transform
        self.questions = self._load_questions()

    def _load_questions(self) -> list[str]:
        with open(self.file_path, encoding="utf-8") as f:
            fields = (line.strip().split("@@") for line in f)
            return [parts[1] for parts in fields if len(parts) >= 2]

    def __len__(self) -> int:
        return len(self.questions)

    def __getitem__(self, idx: int) -> dict:
        enc = self.tokenizer.encode(self.questions[idx], max_length=self.max_length, return_tensors=False)
        item = {name: torch.tensor(enc[name], dtype=torch.long) for name in FIELDS}
        return item if self.transform is None else self.transform(item)


def create_question_dataloaders(file_path: str, tokenizer, batch_size: int = 16, max_length: int = 512,
                                train_ratio: float = 0.9, shuffle: bool = True, num_workers: int = 4,
                                device=None):
    """``(train_loader, val_loader or None)``: ``int(train_ratio * n)`` questions for training, the
    rest for validation, drawn by ``torch.utils.data.random_split`` from the global generator
    (src/dataset.py:164-202).  ``device``: see the module docstring."""
    dataset = QuestionDataset(file_path=file_path, tokenizer=tokenizer, max_length=max_length)
    n_train = int(train_ratio * len(dataset))
    n_val = len(dataset) - n_train
    parts = random_split(dataset, [n_train, n_val]) if n_val > 0 else [dataset]

    def loader(part, shuffled):
        dl = DataLoader(part, batch_size=batch_size, shuffle=shuffled, num_workers=num_workers,
                        collate_fn=collate_batch if device is None else _items)
        return dl if device is None else DevicePrefetcher(dl, device=device)

    train_loader = loader(parts[0], shuffle)
    return train_loader, (loader(parts[1], False) if n_val > 0 else None)
