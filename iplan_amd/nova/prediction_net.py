"""Trajectory-prediction decoder (mirror of nova/prediction_net.py:6-63): parameter containers
with the reference's ``state_dict`` keys; arithmetic in the fused prediction-learning kernels."""

from .behavior_net import DecoderRNN, _ArenaModule  # noqa: F401  (same layer stack, prediction_net.py:6-26)


class Prediction_Decoder(_ArenaModule):
    def __init__(self, input_size, hidden_size, num_layers, output_size, pred_length, dropout=0.5,
                 teacher_forcing_ratio=0.5):
        super().__init__()
        self.pred_length = pred_length
        self.teacher_forcing_ratio = teacher_forcing_ratio
        self.hidden_size = hidden_size
        if hidden_size != 32:
            raise NotImplementedError("prediction decoder kernels are built for attention_dim = 32")
        self.p = dropout
        self.decoder = DecoderRNN(input_size, hidden_size, output_size, num_layers, dropout)

    def forward(self, last_state, teacher_state, hidden, keep=None):
        """last_state [B,N,1,d], teacher_state [B,N,P,d], hidden [B*N,A] -> predicted [B,N,P,d]
        (nova/prediction_net.py:40-63; inference -- training goes through Prediction_policy.learn).  The
        teacher-forcing coin is drawn from numpy's generator once per step exactly as the reference does."""
        import numpy as np
        import torch
        from .. import ops
        B, N, _, d = last_state.shape
        P = self.pred_length
        dev = last_state.device
        arena = self._single(dev)
        coins = [np.random.random() < self.teacher_forcing_ratio for _ in range(P)]
        teacher = torch.as_tensor(np.array([coins], dtype=np.int32), device=dev) if any(coins) else None
        p = self.p if self.training else 0.0
        if keep is None and p > 0:
            keep = torch.empty(1, P, B * N, self.hidden_size, device=dev).bernoulli_(1.0 - p)
        out = ops.pdec_forward(arena, last_state.float().reshape(1, B * N, d).contiguous(),
                               hidden.float().reshape(1, B * N, self.hidden_size).contiguous(),
                               teacher_state.float().reshape(1, B * N, P, d).contiguous(),
                               torch.ones(1, B, device=dev), N, keep=keep, drop_p=p, teacher=teacher)
        return out["pred"][0].reshape(B, N, P, d)
