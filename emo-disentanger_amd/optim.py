"""Fused multi-tensor grad-norm clip + Adam over the flat parameter buffer.

Replaces `torch.nn.utils.clip_grad_norm_(model.parameters(), 0.5); optim.step()`
(/root/reference/stage2_accompaniment/train.py:79-80, optimizer construction :318-326) with three
launches: sum of squares of the flat gradient, clip coefficient (with the 1/world DP pre-scale folded
in), and one Adam kernel that also refreshes the bf16 mirror of the weights.  It is a
``torch.optim.Optimizer`` so `param_groups[0]['lr'] = ...` and `CosineAnnealingLR(optimizer, ...)`
keep working, and its ``state_dict()`` has torch.optim.Adam's layout (per-parameter `step`,
`exp_avg`, `exp_avg_sq` in parameter-registration order) so `trained_optim` checkpoints interchange."""
import torch

from . import ops


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, max_grad_norm=None, world_size=1, token_weighted=False):
        """world_size > 1: the gradient buffer holds the SUM over ranks (dp.allreduce_grads_).  token_weighted=False: every rank
        back-propagated its own mean loss -> pre-scale 1/world.  token_weighted=True: every rank back-propagated the SUM of its
        token losses and the all-reduced non-pad token count sits in the tail slot of the buffer -> pre-scale 1/count, read on
        the device (the exact global mean for unequal token counts)."""
        self.model = model
        params = [p for p in model.parameters() if p.requires_grad]
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=0, amsgrad=False))
        self.max_grad_norm, self.world_size, self.token_weighted = max_grad_norm, world_size, token_weighted
        self._step, self._m, self._v = 0, None, None
        self.last_grad_norm = None

    def _buffers(self):
        ps = self.model._ensure_store()
        if self._m is None or self._m.numel() != ps.total or self._m.device != ps.device:
            self._m = torch.zeros(ps.total, device=ps.device)
            self._v = torch.zeros(ps.total, device=ps.device)
            self._ss = torch.zeros(ops.SUMSQ_FLOATS, device=ps.device)      # [0] = sum of squares, the rest: the kernel's ordered partials
            self._coef = torch.ones(1, device=ps.device)
        return ps

    @torch.no_grad()
    def step(self, closure=None):
        ps = self._buffers()
        ps.ensure_grads()
        from .engine import join_side_stream
        join_side_stream()
        g = self.param_groups[0]
        weighted = self.token_weighted and self.world_size > 1
        if self.world_size > 1:
            # the exchange (dp.allreduce_grads_ / GradExchange.finish) records whether the tail slot carries the all-reduced token
            # count; dividing a token-SUM gradient by the world size, or a mean gradient by an empty tail slot, must not pass silently
            tail = getattr(ps, 'tail_tokens', None)
            if tail is not None and tail != self.token_weighted:
                raise RuntimeError('FusedAdam(token_weighted=%s) but the gradient exchange %s the token count in the tail slot'
                                   % (self.token_weighted, 'carried' if tail else 'did not carry'))
        pre = 1.0 if weighted else 1.0 / self.world_size
        ops.sumsq(ps.flat_grad, self._ss)
        ops.clip_coef(self._ss, float(self.max_grad_norm) if self.max_grad_norm else 3.0e38, pre, self._coef,
                      denom=ps.flat_grad_ext[ps.total:ps.total + 1] if weighted else None)
        self.last_grad_norm = self._ss[0:1]      # device scalar: sqrt(.)*pre is the global grad norm (no host sync here)
        self._step += 1
        ops.adam_step(ps.flat32, ps.flat_grad, self._m, self._v, ps.flat16, g['lr'], g['betas'][0], g['betas'][1], g['eps'], self._step,
                      self._coef)
        ps.mark_mirror_fresh()                   # the kernel rewrote every element of the bf16 mirror from the new fp32 master

    def zero_grad(self, set_to_none=False):
        ps = self.model._ensure_store()
        ps.ensure_grads()
        ps.flat_grad.zero_()

    # ---- torch.optim.Adam-compatible checkpoint layout
    def state_dict(self):
        ps = self._buffers()
        names = [n for n, p in self.model.named_parameters() if p.requires_grad]
        state = {}
        for i, n in enumerate(names):
            o, k = ps.offsets[n], ps.params[n].numel()
            state[i] = {'step': torch.tensor(float(self._step)), 'exp_avg': self._m[o:o + k].view(ps.shapes[n]).clone(),
                        'exp_avg_sq': self._v[o:o + k].view(ps.shapes[n]).clone()}
        grp = {k: v for k, v in self.param_groups[0].items() if k != 'params'}
        grp['params'] = list(range(len(names)))
        return {'state': state, 'param_groups': [grp]}

    def load_state_dict(self, sd):
        ps = self._buffers()
        names = [n for n, p in self.model.named_parameters() if p.requires_grad]
        for i, n in enumerate(names):
            st = sd['state'].get(i)
            if st is None:
                continue
            o, k = ps.offsets[n], ps.params[n].numel()
            self._m[o:o + k].copy_(st['exp_avg'].reshape(-1))
            self._v[o:o + k].copy_(st['exp_avg_sq'].reshape(-1))
            self._step = int(float(st['step']))
        for k, v in sd['param_groups'][0].items():
            if k != 'params':
                self.param_groups[0][k] = v
