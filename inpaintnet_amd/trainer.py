"""Trainer base class: the reference's optimizer loop on the HIP kernels.

Mirrors utils/trainer.py:15-376 of the reference: Trainer(dataset, model, lr,
early_stopping), train_model, loss_and_acc_on_epoch, zero_grad, step, the
abstract loss_and_acc_for_batch / process_batch_data / update_scheduler, and
the static loss helpers.  Differences that are the point of the build:
  * zero_grad() / step() act on the model's flat arenas: one memset, one fused
    Adam kernel (torch.optim.Adam semantics, lr=1e-4, betas (0.9,0.999), eps 1e-8);
  * with torch.distributed initialised (one process per GPU, backend "nccl" =
    RCCL), step() first sums the flat gradient arena over ranks in ONE all-reduce
    and folds the 1/world_size into the Adam kernel;
  * losses are the fused cross-entropy / KL kernels (autograd Functions);
  * loss/accuracy are accumulated on the device, read back once per epoch
    (the reference syncs every batch: utils/trainer.py:154).
Plotting / tensorboard plumbing of the reference is out of scope.
"""
import gc
import os
import sys
import time
from abc import ABC, abstractmethod
from collections import deque

import torch

from . import dp, ops
from .feed import DeviceFeed


def _dist_ready():
    return torch.distributed.is_available() and torch.distributed.is_initialized()


def settle_python_heap():
    """gc.collect() + gc.freeze(): everything alive now (torch's, numpy's and this package's modules: ~170 k tracked objects)
    moves to the permanent generation, which later collections do not traverse.

    Why a trainer cares (round 5, profiles/r05_cold_start.txt): a training step queues ~80 launches in ~1 ms of host time and
    the GPU needs 3.6 ms for them, so the host runs ahead -- until CPython's cyclic collector decides on a FULL collection.
    In a fresh process the third generation's threshold trips within the first dozen steps (the step objects, autograd nodes
    and ctypes arguments are the young objects that count towards it), and one traversal of the import-time heap takes
    36-40 ms on the GPU box: the launch queue drains, the GPU idles for ten steps' worth of time, and a 20-step epoch (a
    validation pass, the driver's `bench.py --steps 20 --warmup 5`) measures 1.7x the steady state -- the whole of round 4's
    "cold-start transient" (6.31 vs 3.65 ms per step).  With the old heap frozen a full collection only looks at what was
    created since: < 1 ms.

    A process-global side effect (INTEGRATION.md section 1 states it): objects alive at the FIRST trainer's construction are never
    examined by the cyclic collector again (reference counting still frees them).  Once per process -- later trainers do not
    freeze what the application has built since; INET_GC_FREEZE=0 leaves the collector alone altogether."""
    global _heap_settled
    if _heap_settled or os.environ.get("INET_GC_FREEZE", "1") == "0":
        return False
    gc.collect()
    gc.freeze()
    _heap_settled = True
    return True


_heap_settled = False


class _CrossEntropyFn(torch.autograd.Function):
    """mean CE over rows + accuracy, one wavefront per row (utils/trainer.py:271-306)."""

    @staticmethod
    def forward(ctx, weights2d, targets1d):
        rows, V = weights2d.shape
        out = torch.zeros(2, dtype=torch.float32, device=weights2d.device)
        need = ctx.needs_input_grad[0]
        dW = torch.empty_like(weights2d) if need else None
        ops.cross_entropy(weights2d, targets1d, out, dW=dW, scale=1.0 / rows, out_scale=1.0 / rows)
        ctx.dW = dW
        loss, acc = out[0], out[1]                  # the kernel accumulated the means: no follow-up launch
        ctx.mark_non_differentiable(acc)
        ctx.set_materialize_grads(False)            # (no fill launch for the accuracy's gradient)
        return loss, acc

    @staticmethod
    def backward(ctx, gloss, _gacc):
        dW = ctx.dW
        ctx.dW = None
        if gloss is None or dW is None:
            return None, None
        return dW * gloss, None


class _ElboFn(torch.autograd.Function):
    """loss = mean CE + kscale * kl_sum, accuracy -- VAETrainer.loss_and_acc_for_batch (vae_trainer.py:29-40) in ONE launch
    forward (the CE kernel also adds the KL term) and ONE launch backward (the same kernel writes dW already multiplied
    by the upstream gradient, read on the device, and hands kscale * gradient on to the KL sum's producer): the dozen
    one-element torch kernels of the unfused expression (zeros, mul, add, select-backward, dW * g) are gone."""

    @staticmethod
    def forward(ctx, weights2d, targets1d, kl_sum, kscale, out2):
        rows, V = weights2d.shape
        ops.cross_entropy_ex(weights2d, targets1d, loss_sum=out2[0:1], correct=out2[1:2], out_scale=1.0 / rows,
                             add_term=kl_sum.reshape(1), add_scale=kscale)
        ctx.save_for_backward(weights2d, targets1d)
        ctx.kscale = kscale
        # (out2 is a slice of the gradient arena's tail, zeroed by the next zero_grad(): the caller gets its own two floats -- the
        #  reference returns independent tensors, and a training script may keep them across steps)
        res = out2.clone()
        loss, acc = res[0], res[1]
        ctx.mark_non_differentiable(acc)
        ctx.set_materialize_grads(False)            # (no fill launch for the accuracy's gradient)
        return loss, acc

    @staticmethod
    def backward(ctx, gloss, _gacc):
        if gloss is None:
            return None, None, None, None, None
        w, t = ctx.saved_tensors
        dW = torch.empty_like(w)
        gkl = torch.empty((), dtype=torch.float32, device=w.device)
        ops.cross_entropy_ex(w, t, dW=dW, scale=1.0 / w.shape[0], scale_dev=gloss.reshape(1).contiguous(),
                             fwd_out=gkl.reshape(1), fwd_scale=ctx.kscale)
        return dW, None, gkl, None, None


class _KLFn(torch.autograd.Function):
    """beta * mean_b sum_d KL(N(mu, sigma) || N(0,1))   (vae_trainer.py:128-139)."""

    @staticmethod
    def forward(ctx, mu, ls, beta):
        acc = torch.zeros(1, dtype=torch.float32, device=mu.device)
        ops.reparam_kl(mu, ls, None, kl_sum=acc)
        ctx.save_for_backward(mu, ls)
        ctx.k = beta / mu.shape[0]
        return acc[0] * ctx.k

    @staticmethod
    def backward(ctx, g):
        mu, ls = ctx.saved_tensors
        dmu, dls = ops.latent_bwd(None, mu, ls, None, ctx.k)
        return dmu * g, dls * g, None


class Trainer(ABC):
    """utils/trainer.py:15-39"""

    feed_fields = (0, 1)       # which members of a (score, metadata) batch the trainer needs on the device

    def __init__(self, dataset, model, lr=1e-4, early_stopping=False):
        self.dataset = dataset
        self.model = model
        self.lr = lr
        self.betas = (0.9, 0.999)
        self.eps = 1e-8
        self.adam_m = torch.zeros_like(model.flat)
        self.adam_v = torch.zeros_like(model.flat)
        self.adam_t = 0
        self.early_stopping = False
        if early_stopping:
            self.early_stopping = True
            self.early_stopper = EarlyStopping()
        self.last_epoch_seconds = None
        self.start_epoch = 0                         # load_training_state() moves it
        self.chain_fallback = True                   # on a chain-kernel timeout: per-step kernels + retry (else raise)
        self.chain_fallbacks = 0
        self.lost_steps = 0                          # optimizer steps the device skipped (their batches were run again)
        # deferred side-stream joins for zero_grad() -> forward -> backward -> step() sequences (see zero_grad)
        self.overlap_backward = False
        # Step reports (inet_adam_step_ex): every optimizer launch leaves a record of what it decided; the host reads the
        # record of step k when it has queued step k + report_lag -- by then that kernel has normally run, so the read costs
        # nothing, and it happens at the SAME step on every rank of a data-parallel job (the decision itself is the ranks'
        # summed flag), so all ranks fall back and repeat the same batches together.
        # The lag is also how far the host may run ahead, and that must be FAR: with a lag of 2 the B = 256 step measured 3.87 ms
        # against 3.63 without any report (4: 3.78, 8: 3.66-3.73, 14: 3.62-3.64; one box, profiles/r04_b_report_lag.txt) whatever the
        # wait was made of (event synchronize, event query, a plain host spin on the pinned word): the host's launch loop has
        # pauses of several milliseconds now and then, and only a deep queue hides them from the GPU.
        self.report_lag = int(os.environ.get("INET_REPORT_LAG", "12"))
        self._tag = 0
        self._inflight = deque()                     # tags whose report has not been read yet, oldest first
        self._reports = None                         # ring of _NREP records in pinned host memory, written by the kernel
        self._report_events = None                   # one event per record, recorded behind its optimizer launch
        self._recent = deque(maxlen=self._NREP)      # (tag, batch) of the epoch loop's last steps: what a fallback runs again
        self._lost = []                              # tags found skipped since the last fallback
        # the host must stay ahead of the GPU from the first step on: no full garbage collection over the import-time heap
        # in the middle of the first steps (settle_python_heap)
        settle_python_heap()
        if model.grad.is_cuda:                       # the report ring and its events exist before the first step, not in it
            # ... and so does every kernel of the library: the HIP runtime loads code objects and builds function objects on the
            # launch path of a kernel's FIRST launch, and which kernels a step meets first depends on its branch (the first
            # free-running step of a process paid 1.3 ms of host time for it, 11-16 ms now and then: csrc/preload.hip)
            ops.preload()
            self._report_slot(0)
            for e in self._report_events:
                e.record()

    # ---- utils/trainer.py:41-124 (plot/log plumbing omitted) -----------------------
    def train_model(self, batch_size, num_epochs, plot=False, log=False, seed=0):
        """Epoch loop of the reference (train pass, validation pass, stats, save, optional early stopping).
        `batch_size` is the GLOBAL batch.  Under torch.distributed every rank iterates the same loader (same shuffle:
        the host generators are seeded identically), takes its contiguous shard of every batch, and sees the same
        epoch statistics (they are summed over ranks), so save / early-stopping decisions agree on all ranks."""
        if dp.world_size() > 1:
            dp.seed_shared(seed)
            dp.seed_rank(seed)
            dp.broadcast_params(self.model.flat)
        train_loader, val_loader, _ = self.dataset.data_loaders(batch_size=batch_size, split=(0.70, 0.20))
        print('Num Train Batches: ', len(train_loader))
        print('Num Valid Batches: ', len(val_loader))
        try:
            for epoch in range(self.start_epoch, num_epochs):
                self.update_scheduler(epoch)
                stats = []
                for loader, train in ((train_loader, True), (val_loader, False)):
                    self.model.train(train)
                    stats += self.loss_and_acc_on_epoch(data_loader=loader, epoch_num=epoch, train=train)
                self.print_epoch_stats(epoch, num_epochs, *stats)
                if dp.rank() == 0:
                    self.model.save()
                    if epoch > 0 and epoch % 10 == 0:
                        self.model.save_checkpoint(epoch)
                        self.save_training_state(self.model.filepath + f'_{epoch}.trainer', epoch + 1)
                if self.early_stopping and self.early_stopper(stats[2], self.model):
                    print("Early Stopping")
                    return
        finally:
            if self._inflight:                       # (every exit path: nothing a late report holds may go unread)
                if sys.exc_info()[0] is None:
                    self.finish()
                else:                                # an exception is on its way out already: do not mask it (as __exit__)
                    try:
                        self.finish()
                    except Exception:
                        pass

    # ---- utils/trainer.py:126-163 ---------------------------------------------------
    def loss_and_acc_on_epoch(self, data_loader, epoch_num=None, train=True):
        dev = self.model.flat.device
        sums = torch.zeros(5, dtype=torch.float32, device=dev)        # loss sum, accuracy sum, batches, chain timeouts, bad tokens
        t0 = time.time()
        prev_overlap, self.overlap_backward = self.overlap_backward, bool(train)
        try:
            if not isinstance(data_loader, DeviceFeed):
                # this rank's shard of every batch, copied on a side stream two batches ahead (feed.py)
                data_loader = DeviceFeed(data_loader, fields=self.feed_fields, device=dev)
            for sample_id, batch in enumerate(data_loader):
                batch_data = self.process_batch_data(batch)
                self._run_batch(batch_data, epoch_num, train, sums)
            if train:                                # the reports of the last report_lag steps (replays what they lost)
                self._settle(epoch_num, sums)
        finally:
            self.overlap_backward = prev_overlap
            ops.side_defer(False)
            self._recent.clear()
        # Chain launches outside optimizer steps (validation passes) have no step report: their status travels with the sums, so
        # that every rank sees every rank's failures and all of them raise together instead of one leaving the others in a
        # collective.  (One synchronisation per epoch; the sums' read-back was one already.)
        if sums.is_cuda:
            torch.cuda.synchronize(dev)
            sums[3] = float(max(ops.chain_status(), 0))
            sums[4] = float(max(ops.token_status(), 0))     # (validation passes have no step report: the count travels here)
        dp.allreduce_sum_(sums)                      # every rank reports (and early-stops on) the global means
        out = sums.tolist()                          # the one device->host sync of the epoch
        if sums.is_cuda:
            ops.chain_status(reset=True)             # persistent kernels: bounded spins report here instead of hanging
            ops.token_status(reset=True)
        if out[4] > 0:                               # on every rank together (decoder.py:36-45 check_index)
            raise ops.TokenRangeError(f"Invalid Value of index: {int(out[4])} launch(es) (all ranks) met a token outside "
                                      "[0, num_notes) (Trainer.loss_and_acc_on_epoch); the results computed from it are not valid")
        if out[3] > 0:
            raise RuntimeError(f"{int(out[3])} chain-kernel workgroups (all ranks) gave up waiting for their group during "
                               "this epoch outside an optimizer step (inet_chain_status); its results are not valid")
        n = max(out[2], 1.0)
        self.last_epoch_seconds = time.time() - t0
        return out[0] / n, out[1] / n

    def _run_batch(self, batch_data, epoch_num, train, sums, replay=False):
        """One batch of the epoch loop.  A ChainTimeoutError out of step() means: an optimizer step issued report_lag steps ago
        (or later) skipped itself because a persistent kernel of some rank gave up waiting for its group.  Nothing is corrupted
        -- on EVERY rank the optimizer kernel has left the weights alone since (it reads the ranks' summed flag) and the masked
        epoch statistics left those batches out -- so all ranks switch to the per-step kernels and run the lost batches again,
        in order."""
        try:
            self.zero_grad()
            if train:
                loss, accuracy = self.loss_and_acc_for_batch(batch_data, epoch_num, train=True)
                loss.backward()
                self._recent.append((self._tag, batch_data))       # step() issues its optimizer launch under this tag
                self.step()
            else:
                with torch.no_grad():
                    loss, accuracy = self.loss_and_acc_for_batch(batch_data, epoch_num, train=False)
        except ops.ChainTimeoutError:
            if replay or not self.chain_fallback:
                raise
            self._replay(self._fall_back_from_chains(), epoch_num, sums)
            return
        self._accumulate_stats(sums, loss, accuracy, self._flag())

    def _replay(self, lost, epoch_num, sums):
        batches = dict(self._recent)
        for tag in lost:
            if tag not in batches:
                raise RuntimeError(f"optimizer step {tag} was skipped after a chain-kernel timeout and its batch is no longer "
                                   "held (report_lag too large for the replay window)")
            self._run_batch(batches[tag], epoch_num, True, sums, replay=True)
        self._settle(epoch_num, sums, replay=True)

    def _settle(self, epoch_num, sums, replay=False):
        """Read every outstanding step report (waits for those optimizer launches); lost steps are run again."""
        try:
            self.check_steps(wait_all=True)
        except ops.ChainTimeoutError:
            if replay or not self.chain_fallback:
                raise
            self._replay(self._fall_back_from_chains(), epoch_num, sums)

    def _flag(self):
        """The device word that decides whether this step counts: the ranks' summed chain status under data parallelism
        (None = the process's own status word)."""
        return self.model.step_flag if dp.world_size() > 1 and self._reports_on() else None

    @staticmethod
    def _accumulate_stats(sums, loss, accuracy, step_flag=None):
        """sums[:3] += (loss, accuracy, 1) on the device in one launch (inet_epoch_stats_add_ex); a batch whose chain kernels
        timed out on any rank -- its results are not valid and the optimizer kernel skipped it -- stays out of the means."""
        loss = loss.detach()
        if loss.dim() > 0:
            loss = loss.mean()
        loss = loss.reshape(1).to(sums.dtype).contiguous()
        acc = None if accuracy is None else accuracy.detach().reshape(1).to(sums.dtype).contiguous()
        if not sums.is_cuda:                         # (the world_size-2 gloo tests drive this loop with host tensors)
            if step_flag is not None and float(step_flag[0]) != 0.0:
                return
            sums[0] += loss[0]
            if acc is not None:
                sums[1] += acc[0]
            sums[2] += 1.0
            return
        ops.epoch_stats_add(sums, loss, acc, step_flag)

    def zero_grad(self):
        """utils/trainer.py:165-170.  With `overlap_backward` set (the epoch loop and bench.py set it) the step that
        starts here runs with deferred side-stream joins: the gradient arena is only complete after step() (or
        ops.side_join()), not right after loss.backward()."""
        dp.reset_buckets(self.model.grad)            # nothing may survive from a step that never reached step()
        self.model.__dict__.pop("_dp_open", None)    # (nor counts of forward passes that never saw their backward)
        ops.side_defer(bool(self.overlap_backward))
        self.model.zero_grad()

    def step(self):
        """utils/trainer.py:172-177 (+ the data-parallel gradient exchange).  May raise ops.ChainTimeoutError / ValueError
        for an EARLIER step (see check_steps)."""
        self._join_side_work()
        m = self.model
        flag = self._flag()
        if flag is not None:
            self._export_flag(flag)                  # this rank's chain status -> the word in front of the arena ...
        gscale = dp.allreduce_grads(m.grad, store=m._grad_store, head=m._HEAD)     # ... summed with the gradients
        ops.release_held()                           # (after the exchange: the buckets' streams were ordered behind side work)
        self.adam_t += 1
        tag = self._tag
        self._tag += 1
        if self._reports_on():
            if len(self._inflight) >= self._NREP - 1:     # (only if check_steps() was overridden away: never reuse an unread record)
                self.check_steps(wait_all=True)
            self._launch_optimizer(tag, gscale, flag)
            self._inflight.append(tag)
            self.check_steps()
        else:
            ops.adam_step(m.flat, m.grad, self.adam_m, self.adam_v, self.lr, self.adam_t, self.betas[0], self.betas[1],
                          self.eps, gscale)

    # ---- the device-touching pieces of the protocol, one method each: tests/test_dp_gloo.py drives step(), check_steps() and the
    # ---- fallback / replay logic with host stand-ins for them (world 2, gloo)
    def _reports_on(self):
        return self.model.grad.is_cuda

    def _join_side_work(self):
        ops.side_defer(False, release=False)         # joins; deferred mode only lives between zero_grad() and step()

    def _export_flag(self, flag):
        ops.step_flag_export(flag)

    def _launch_optimizer(self, tag, gscale, flag):
        m = self.model
        rec = self._report_slot(tag)
        rec.zero_()                                  # host write: the launch that last wrote this record was waited for when it was read
        ops.adam_step(m.flat, m.grad, self.adam_m, self.adam_v, self.lr, self.adam_t, self.betas[0], self.betas[1],
                      self.eps, gscale, step_flag=flag, report=rec)
        self._report_events[tag % self._NREP].record()

    def _device_sync(self):
        torch.cuda.synchronize()

    def _chains_off(self):
        """Clear the status words and switch this process to the per-step kernels; returns this rank's timed-out workgroups."""
        n = ops.chain_status(reset=True)
        ops.set_option(4, 0)                          # per-step kernels from here on (INET_CHAIN=0 semantics, in-process)
        return n

    _NREP = 16

    def _report_slot(self, tag):
        if self._reports is None:
            self._reports = torch.zeros(self._NREP, 4, dtype=torch.int32).pin_memory()
            self._report_events = [torch.cuda.Event() for _ in range(self._NREP)]
        return self._reports[tag % self._NREP]

    def _read_report(self, tag):
        """(executed, skipped, nonfinite) of the optimizer launch issued under `tag`, after the event behind it."""
        self._report_events[tag % self._NREP].synchronize()
        r = self._reports[tag % self._NREP].tolist()
        if not r[0]:
            raise RuntimeError(f"optimizer launch {tag} left no report (the kernel did not run?)")
        return bool(r[0]), bool(r[1]), bool(r[2]), bool(r[3])

    def check_steps(self, wait_all=False):
        """Read the reports of the optimizer steps issued at least `report_lag` steps ago (all outstanding ones with
        wait_all).  Raises ops.ChainTimeoutError if one of them skipped itself -- a persistent kernel of SOME rank had timed out,
        see _run_batch --, ValueError if a parameter became NaN / inf in one of them (the reference's "... has become nan",
        MeasureVAE/encoder.py:111-116, decoder.py:424-429), ops.TokenRangeError for a token outside the vocabulary
        (decoder.py:36-45).  Identical on every rank of a data-parallel job: same reports, read at the same step."""
        newest = self._tag - 1
        skipped = nonfinite = badtok = False
        while self._inflight and (wait_all or self._inflight[0] <= newest - self.report_lag):
            tag = self._inflight.popleft()
            rep = self._read_report(tag)
            skip, bad, tok = rep[1], rep[2], (rep[3] if len(rep) > 3 else False)
            if skip:
                self._lost.append(tag)
            skipped |= skip
            nonfinite |= bad
            badtok |= tok
        if skipped:
            raise ops.ChainTimeoutError(
                f"optimizer step(s) {self._lost} skipped: a chain kernel gave up waiting for its group (on this or another "
                "rank), the gradients of those steps are not valid and the weights were left alone.  All workgroups of such a "
                "launch must be resident at once -- is the GPU shared, partitioned or CU-masked?  INET_CHAIN=0 (or "
                f"ops.set_option(4, 0)) selects the per-step kernels.  Recorder: {ops.slow_waits_summary()}")
        if nonfinite:
            raise ValueError(f"{type(self.model).__name__} has become nan (a parameter left the finite range in an optimizer step)")
        if badtok:
            # data parallel: SOME rank's prologue kernels met a token outside the vocabulary (the token words travel with the
            # step flag and come back in the report): every rank raises here, at the same step -- nobody is left in a collective
            if self.model.flat.is_cuda:
                ops.token_status(reset=True)
            raise ops.TokenRangeError("Invalid Value of index: a token outside [0, num_notes) reached the model on this or another "
                                      "rank (Trainer.step); the results computed from it are not valid")
        if self.model.flat.is_cuda and self._flag() is None:
            ops.check_tokens("Trainer.step")         # single process: the host-mapped counter of this process

    def finish(self):
        """Read every outstanding step report NOW (waits for the optimizer launches still in flight) and raise what they hold.
        The errors of a step surface up to `report_lag` steps late by design (the host must run ahead of the GPU); a manual
        training loop -- zero_grad / loss / backward / step without loss_and_acc_on_epoch -- calls this after its last step,
        or runs inside `with trainer:`, so that a NaN weight, a bad token or a chain timeout in the LAST steps is not lost.
        The epoch loop and train_model do it themselves."""
        self.check_steps(wait_all=True)

    close = finish

    def __enter__(self):
        return self

    def __exit__(self, exc_type, exc, tb):
        if exc_type is None:
            self.finish()
        else:                                        # do not mask the exception that is already on its way out
            try:
                self.finish()
            except Exception:
                pass
        return False

    def _fall_back_from_chains(self):
        """After a ChainTimeoutError out of step() / check_steps(): wait for the device, find every skipped step, take their
        count back out of the bias-correction step number, clear the status words, switch this process to the per-step
        kernels.  Returns the tags of the lost steps, oldest first (every rank computes the same list)."""
        self._device_sync()
        while self._inflight:
            tag = self._inflight.popleft()
            if self._read_report(tag)[1]:
                self._lost.append(tag)
        lost, self._lost = sorted(set(self._lost)), []
        n = self._chains_off()
        self.chain_fallbacks += 1
        self.lost_steps += len(lost)
        self.adam_t = max(self.adam_t - len(lost), 0)  # those launches did not update anything
        print(f"[inpaintnet_amd] rank {dp.rank()}: {max(n, 0)} chain-kernel workgroups of this rank timed out; optimizer steps "
              f"{lost} were skipped on every rank; chain kernels are now OFF for this process and those batches are run again "
              "(is the GPU shared, partitioned or CU-masked?)")
        return lost

    # ---- optimizer / epoch resume (SURVEY 8f1 add-on: the reference saves weights only, utils/model.py:16-53) ----
    def training_state(self, next_epoch=0):
        return {"adam_m": self.adam_m.cpu(), "adam_v": self.adam_v.cpu(), "adam_t": self.adam_t, "lr": self.lr,
                "betas": self.betas, "eps": self.eps, "next_epoch": int(next_epoch),
                "num_parameters": int(self.model.flat.numel())}

    def save_training_state(self, path, next_epoch=0):
        torch.save(self.training_state(next_epoch), path)

    def load_training_state(self, path_or_state):
        """Restore the Adam moments, step count and the epoch to resume from (train_model starts there)."""
        st = torch.load(path_or_state, map_location="cpu") if isinstance(path_or_state, (str, bytes, os.PathLike)) \
            else path_or_state
        if int(st["num_parameters"]) != self.model.flat.numel():
            raise RuntimeError("training state belongs to a model with a different parameter arena")
        self.adam_m.copy_(st["adam_m"])
        self.adam_v.copy_(st["adam_v"])
        self.adam_t = int(st["adam_t"])
        self.lr, self.betas, self.eps = float(st["lr"]), tuple(st["betas"]), float(st["eps"])
        self.start_epoch = int(st["next_epoch"])
        return self.start_epoch

    @abstractmethod
    def loss_and_acc_for_batch(self, batch, epoch_num=None, train=True):
        pass

    @abstractmethod
    def process_batch_data(self, batch):
        pass

    @abstractmethod
    def update_scheduler(self, epoch_num):
        pass

    @staticmethod
    def print_epoch_stats(epoch_index, num_epochs, mean_loss_train, mean_accuracy_train, mean_loss_val,
                          mean_accuracy_val):
        print(f'Train Epoch: {epoch_index + 1}/{num_epochs}')
        print(f'\tTrain Loss: {mean_loss_train}\tTrain Accuracy: {mean_accuracy_train * 100} %')
        print(f'\tValid Loss: {mean_loss_val}\tValid Accuracy: {mean_accuracy_val * 100} %')

    # ---- static losses, utils/trainer.py:271-376 -------------------------------------
    @staticmethod
    def mean_crossentropy_loss_and_accuracy(weights, targets):
        V = weights.size(-1)
        w2 = weights.contiguous().view(-1, V)
        t1 = targets.contiguous().view(-1)
        if t1.dtype != torch.int64:
            t1 = t1.long()
        return _CrossEntropyFn.apply(w2, t1)

    @staticmethod
    def mean_crossentropy_loss(weights, targets):
        """weights (B,T,V), targets (B,T)   (utils/trainer.py:271-288)"""
        batch_size, seq_len, num_notes = weights.size()
        assert batch_size == targets.size(0)
        assert seq_len == targets.size(1)
        return Trainer.mean_crossentropy_loss_and_accuracy(weights, targets)[0]

    @staticmethod
    def mean_accuracy(weights, targets):
        """utils/trainer.py:290-306 (argmax = lowest index among maxima, as Tensor.max(1))"""
        with torch.no_grad():
            return Trainer.mean_crossentropy_loss_and_accuracy(weights.detach(), targets)[1]

    @staticmethod
    def mean_crossentropy_loss_alt(weights, targets):
        """weights (B,M,T,V), targets (B,M,T)   (utils/trainer.py:344-359)"""
        return Trainer.mean_crossentropy_loss_and_accuracy(weights, targets)[0]

    @staticmethod
    def mean_accuracy_alt(weights, targets):
        """utils/trainer.py:361-376"""
        with torch.no_grad():
            return Trainer.mean_crossentropy_loss_and_accuracy(weights.detach(), targets)[1]

    @staticmethod
    def mean_mse_loss_rnn(weights, targets):
        """utils/trainer.py:327-342 (diagnostic only: never part of a training loss in the reference)"""
        assert weights.size() == targets.size()
        return ((weights - targets) ** 2).mean()

    @staticmethod
    def mean_l1_loss_rnn(weights, targets):
        """utils/trainer.py:308-325"""
        assert weights.size() == targets.size()
        return (weights - targets).abs().mean()


class EarlyStopping:
    """Patience counter on the validation loss (behaviour of utils/trainer.py:379-413): an epoch counts as an
    improvement only if the loss drops by at least `min_delta` below the best seen; `patience` epochs in a row without
    one set `early_stop`.  Calling the object returns the flag."""

    def __init__(self, patience=5, verbose=False, min_delta=1e-5):
        self.patience = patience
        self.verbose = verbose
        self.min_delta = min_delta
        self.counter = 0
        self.best_score = None
        self.early_stop = False
        self.val_loss_min = float("inf")

    def __call__(self, val_loss, model=None):
        score = -float(val_loss)
        if self.best_score is None or score - self.best_score >= self.min_delta:
            if self.best_score is not None:
                self.counter = 0
                self.val_loss_min = float(val_loss)
            self.best_score = score
        else:
            self.counter += 1
            self.early_stop = self.early_stop or self.counter >= self.patience
        return self.early_stop
