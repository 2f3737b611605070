"""Training-step driver with the reference trainer's contract.

Mirrors ``DiffusionLanguageModelTrainer`` (src/trainer.py:12-193): constructor
``(model, optimizer, scheduler, accelerator)``, ``train_epoch`` and ``train``
with the same step order (zero_grad -> model -> variational_lower_bound_loss ->
accelerator.backward -> optimizer.step -> loss.item()), ``history`` and
``global_step``.  ``evaluate`` reports loss and masked-token accuracy; the
sampling / decoding examples of the reference's evaluate are not reproduced.
"""

from __future__ import annotations

import torch

from .loss import variational_lower_bound_loss


class DiffusionLanguageModelTrainer:
    def __init__(self, model, optimizer, scheduler, accelerator):
        self.model = model
        self.optimizer = optimizer
        self.scheduler = scheduler
        self.accelerator = accelerator
        self.history: list[float] = []
        self.global_step = 0
        self.fused_head_loss = True   # False: model() then variational_lower_bound_loss(), as the reference spells it

    def _step(self, batch) -> torch.Tensor:
        input_ids = batch["input_ids"]
        attention_mask = batch.get("attention_mask", None)
        self.optimizer.zero_grad()
        if self.fused_head_loss and hasattr(self.model, "training_loss"):
            # same value and gradients as the two calls below; the logits of positions the loss
            # never reads are not formed
            loss = self.model.training_loss(input_ids=input_ids, attention_mask=attention_mask)
        else:
            logits, masked_input_ids, _ = self.model(input_ids=input_ids, attention_mask=attention_mask)
            loss = variational_lower_bound_loss(logits=logits, targets=input_ids,
                                                masked_input_ids=masked_input_ids, attention_mask=attention_mask,
                                                mask_token_id=self.model.tokenizer.mask_token_id)
        self.accelerator.backward(loss)
        self.optimizer.step()
        self.global_step += 1
        return loss

    def train_epoch(self, dataloader) -> float:
        self.model.train()
        total_loss = 0.0
        for batch in dataloader:
            loss = self._step(batch)
            value = loss.item()
            total_loss += value
            print(f"Step {self.global_step} Loss: {value:.4f}")
        epoch_loss = total_loss / len(dataloader)
        self.history.append(epoch_loss)
        if self.scheduler is not None:
            self.scheduler.step()
        current_lr = self.optimizer.param_groups[0]["lr"]
        print(f"Epoch {len(self.history)} Loss: {epoch_loss:.4f}, LR: {current_lr:.8f}")
        return epoch_loss

    def train(self, train_dataloader, val_dataloader=None, num_epochs: int = 1, eval_steps: int = 100,
              eval_mask_ratio: float = 0.15, num_examples: int = 3, sample_steps: int = 10) -> list[float]:
        for epoch in range(num_epochs):
            self.model.train()
            total_loss = 0.0
            for batch_idx, batch in enumerate(train_dataloader):
                loss = self._step(batch)
                value = loss.item()
                total_loss += value
                if batch_idx % 10 == 0:
                    print(f"Epoch {epoch + 1}/{num_epochs} | Batch {batch_idx}/{len(train_dataloader)} | "
                          f"Loss: {value:.4f}")
                if val_dataloader is not None and self.global_step % eval_steps == 0:
                    res = self.evaluate(val_dataloader, mask_ratio=eval_mask_ratio)
                    print(f"\nEvaluation at step {self.global_step}:")
                    print(f"Validation Loss: {res['loss']:.4f}")
                    print(f"Mask Prediction Accuracy: {res['accuracy']:.4f}")
                    self.model.train()
            epoch_loss = total_loss / len(train_dataloader)
            self.history.append(epoch_loss)
            if self.scheduler is not None:
                self.scheduler.step()
            current_lr = self.optimizer.param_groups[0]["lr"]
            print(f"Epoch {epoch + 1}/{num_epochs} | Loss: {epoch_loss:.4f} | LR: {current_lr:.8f}")
            if val_dataloader is not None:
                res = self.evaluate(val_dataloader, mask_ratio=eval_mask_ratio)
                print(f"Epoch {epoch + 1} Validation Loss: {res['loss']:.4f}")
                print(f"Epoch {epoch + 1} Mask Prediction Accuracy: {res['accuracy']:.4f}")
        return self.history

    def evaluate(self, dataloader, num_examples: int = 0, mask_ratio: float = 0.15, sample_steps: int = 10,
                 temperature: float = 0.8) -> dict:
        """Fixed-t validation loss and masked-token accuracy (src/trainer.py:220-255)."""
        self.model.eval()
        total_loss, total_correct, total_masked = 0.0, 0, 0
        with torch.no_grad():
            for batch in dataloader:
                input_ids = batch["input_ids"]
                attention_mask = batch.get("attention_mask", None)
                t = torch.ones(1, device=input_ids.device) * mask_ratio
                logits, masked_input_ids, _ = self.model.forward(input_ids=input_ids,
                                                                 attention_mask=attention_mask, t=t)
                loss = variational_lower_bound_loss(logits=logits, targets=input_ids,
                                                    masked_input_ids=masked_input_ids,
                                                    attention_mask=attention_mask,
                                                    mask_token_id=self.model.tokenizer.mask_token_id)
                total_loss += loss.item()
                predictions = logits.argmax(dim=-1)
                pos = masked_input_ids == self.model.tokenizer.mask_token_id
                if attention_mask is not None:
                    pos = pos & (attention_mask == 1)
                total_correct += (predictions[pos] == input_ids[pos]).sum().item()
                total_masked += pos.sum().item()
        return {"loss": total_loss / max(len(dataloader), 1),
                "accuracy": total_correct / total_masked if total_masked > 0 else 0, "examples": []}
