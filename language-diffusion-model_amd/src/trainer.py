"""Training-step driver with the reference trainer's contract.

Mirrors ``DiffusionLanguageModelTrainer`` (src/trainer.py:12-193): constructor
``(model, optimizer, scheduler, accelerator)``, ``train_epoch`` and ``train``
with the same step order (zero_grad -> model -> variational_lower_bound_loss ->
accelerator.backward -> optimizer.step -> loss.item()), ``history`` and
``global_step``.  ``evaluate`` reports loss, masked-token accuracy and decoded
examples (src/trainer.py:195-319).
"""

from __future__ import annotations

import gc

import torch

from . import ops
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
        self._loss_host = None        # pinned float the step's loss is copied to as soon as it exists
        self._loss_event = None
        self._staged = None
        self._gc_frozen = False
        self.objective = "reference"  # "paper": per-sequence t ~ U(eps,1), 1/t weights, B*L normaliser (not the reference's)
        self.objective_eps = 1e-3
        # Does the optimizer keep the packed compute-dtype weight copies current by itself?  Only
        # src.optim.FusedAdamW attached to the model does.  Anything else -- torch optimizers (the fused
        # ones update parameters without moving their version counters), wrappers such as accelerate's
        # AcceleratedOptimizer (a torch Optimizer subclass that never runs Optimizer.__init__, so it has
        # no hook registry and fires no hooks) -- gets the copies marked stale right after its step().
        from .optim import FusedAdamW
        inner = getattr(optimizer, "optimizer", optimizer)    # accelerate keeps the real one here
        self._marks_stale = not (isinstance(inner, FusedAdamW) and getattr(inner, "_module", None) is not None)
        # One process: a step whose loss had no masked position must change nothing, as in the reference
        # (src/loss.py:18-19 returns a constant, so optimizer.step() at src/trainer.py:75 sees no gradient).
        # FusedAdamW decides that on the device.  With several processes the all-reduce averages that
        # rank's zeros with the other ranks' gradients and every rank steps, which is what DDP does.
        self._gated = inner if (isinstance(inner, FusedAdamW)
                                and int(getattr(accelerator, "num_processes", 1)) == 1) else None

    def _step(self, batch) -> torch.Tensor:
        input_ids = batch["input_ids"]
        attention_mask = batch.get("attention_mask", None)
        self.optimizer.zero_grad()
        if self.fused_head_loss and hasattr(self.model, "training_loss"):
            # same value and gradients as the two calls below; the logits of positions the loss
            # never reads are not formed
            if self.objective == "reference":
                loss = self.model.training_loss(input_ids=input_ids, attention_mask=attention_mask)
            else:
                loss = self.model.training_loss(input_ids=input_ids, attention_mask=attention_mask,
                                                objective=self.objective, eps=self.objective_eps)
        else:
            logits, masked_input_ids, _ = self.model(input_ids=input_ids, attention_mask=attention_mask)
            loss = variational_lower_bound_loss(logits=logits, targets=input_ids,
                                                masked_input_ids=masked_input_ids, attention_mask=attention_mask,
                                                mask_token_id=self.model.tokenizer.mask_token_id)
        self._stage_loss(loss)
        self.accelerator.backward(loss)
        if self._gated is not None and loss.is_cuda:
            self._gated.gate(ops.LAST_LOSS_COUNT)
        self.optimizer.step()
        if self._marks_stale:
            ops.parameters_changed()
        self.global_step += 1
        if not self._gc_frozen:
            # after the first step everything long-lived exists: park it in the collector's permanent
            # generation, so that a later full collection does not pause the host (~80 ms with torch
            # loaded) while the GPU queue runs dry
            gc.collect()
            gc.freeze()
            self._gc_frozen = True
        return loss

    def _stage_loss(self, loss: torch.Tensor) -> None:
        """Queue the device-to-host copy of the loss value right behind the kernels that produce it,
        i.e. BEFORE backward and the optimizer step are enqueued."""
        if not loss.is_cuda:
            self._staged = None
            return
        if self._loss_host is None:
            self._loss_host = torch.empty(1, dtype=torch.float32, pin_memory=True)
            self._loss_event = torch.cuda.Event()
        self._loss_host.copy_(loss.detach().reshape(1), non_blocking=True)
        self._loss_event.record()
        self._staged = loss

    def loss_value(self, loss: torch.Tensor) -> float:
        """``loss.item()`` for the loss ``_step`` just returned (src/trainer.py:78,139 of the reference).
        The value was copied to pinned host memory when it was produced, so the host waits for that
        copy only, not for the backward pass and the optimizer step queued behind it: the GPU still
        has work while the host prepares the next step."""
        if self._staged is loss and loss.is_cuda:
            self._loss_event.synchronize()
            return float(self._loss_host[0])
        return loss.item()

    def train_epoch(self, dataloader) -> float:
        self.model.train()
        total_loss = 0.0
        for batch in dataloader:
            loss = self._step(batch)
            value = self.loss_value(loss)
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
                value = self.loss_value(loss)
                total_loss += value
                if batch_idx % 10 == 0:
                    print(f"Epoch {epoch + 1}/{num_epochs} | Batch {batch_idx}/{len(train_dataloader)} | "
                          f"Loss: {value:.4f}")
                if val_dataloader is not None and self.global_step % eval_steps == 0:
                    res = self.evaluate(val_dataloader, num_examples=num_examples, mask_ratio=eval_mask_ratio,
                                        sample_steps=sample_steps)
                    print(f"\nEvaluation at step {self.global_step}:")
                    print(f"Validation Loss: {res['loss']:.4f}")
                    print(f"Mask Prediction Accuracy: {res['accuracy']:.4f}")
                    print("\nGeneration Examples:")
                    for i, example in enumerate(res["examples"]):
                        print(f"\nExample {i + 1}:")
                        print(f"Original: {example['original']}")
                        print(f"Masked  : {example['masked']}")
                        print(f"Generated: {example['generated']}")
                        print(f"Predicted: {example['predicted']}")
                    print("\n")
                    self.model.train()
            epoch_loss = total_loss / len(train_dataloader)
            self.history.append(epoch_loss)
            if self.scheduler is not None:
                self.scheduler.step()
            current_lr = self.optimizer.param_groups[0]["lr"]
            print(f"Epoch {epoch + 1}/{num_epochs} | Loss: {epoch_loss:.4f} | LR: {current_lr:.8f}")
            if val_dataloader is not None:
                res = self.evaluate(val_dataloader, num_examples=num_examples, mask_ratio=eval_mask_ratio,
                                    sample_steps=sample_steps)
                print(f"Epoch {epoch + 1} Validation Loss: {res['loss']:.4f}")
                print(f"Epoch {epoch + 1} Mask Prediction Accuracy: {res['accuracy']:.4f}")
        return self.history

    def evaluate(self, dataloader, num_examples: int = 3, mask_ratio: float = 0.15, sample_steps: int = 10,
                 temperature: float = 0.8) -> dict:
        """Fixed-t validation loss, masked-token accuracy and up to ``num_examples`` decoded
        examples (first sequence of a batch: original, masked, predicted, and a sample() run
        from its first five tokens) -- src/trainer.py:195-319.  The per-position argmax is one
        kernel over the logits; nothing but the loss and two counters is read back per batch."""
        self.model.eval()
        tok = self.model.tokenizer
        mask_id = tok.mask_token_id
        total_loss, total_correct, total_masked = 0.0, 0, 0
        ex = []
        from . import _native as N
        with torch.no_grad():
            for i, batch in enumerate(dataloader):
                input_ids = batch["input_ids"]
                attention_mask = batch.get("attention_mask", None)
                t = torch.ones(1, device=input_ids.device) * mask_ratio
                logits, masked_input_ids, _ = self.model.forward(input_ids=input_ids,
                                                                 attention_mask=attention_mask, t=t)
                loss = variational_lower_bound_loss(logits=logits, targets=input_ids,
                                                    masked_input_ids=masked_input_ids,
                                                    attention_mask=attention_mask, mask_token_id=mask_id)
                total_loss += loss.item()
                V = logits.shape[-1]
                l2 = logits.reshape(-1, V)
                mids = masked_input_ids.contiguous()
                predictions = torch.empty_like(mids)
                N.check(N.lib().ldm_sample_tokens(l2.data_ptr(), N.dtype_code(l2.dtype), mids.numel(), V,
                                                  l2.stride(0), mids.data_ptr(), mask_id, 1, 0.0, 0, 0, None,
                                                  predictions.data_ptr(), N.stream()), "ldm_sample_tokens")
                pos = masked_input_ids == mask_id
                if attention_mask is not None:
                    pos = pos & (attention_mask == 1)
                total_correct += ((predictions == input_ids) & pos).sum().item()
                total_masked += pos.sum().item()
                if i < num_examples and i < len(dataloader):
                    prompt = input_ids[0:1, :min(5, input_ids.size(1))]
                    generated = self.model.sample(prompt=prompt, num_steps=sample_steps,
                                                  seq_length=input_ids.size(1), temperature=temperature)
                    ex.append((input_ids[0].cpu().tolist(), masked_input_ids[0].cpu().tolist(),
                               predictions[0].cpu().tolist(), generated[0].cpu().tolist()))
        examples = []
        for input_seq, masked_seq, pred_seq, generated_seq in ex[:num_examples]:
            combined = [p if m == mask_id else m for m, p in zip(masked_seq, pred_seq)]
            examples.append({
                "original": tok.decode(input_seq, skip_special_tokens=True),
                "masked": tok.decode(masked_seq, skip_special_tokens=False),
                "predicted": tok.decode(combined, skip_special_tokens=True),
                "generated": tok.decode(generated_seq, skip_special_tokens=True) if generated_seq
                else "No generation available",
            })
        return {"loss": total_loss / len(dataloader),
                "accuracy": total_correct / total_masked if total_masked > 0 else 0, "examples": examples}
