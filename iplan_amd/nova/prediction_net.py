"""Trajectory-prediction decoder (mirror of nova/prediction_net.py:6-63): parameter containers
with the reference's ``state_dict`` keys; arithmetic in the fused prediction-learning kernels."""
import torch.nn as nn

from .behavior_net import DecoderRNN, _ArenaModule  # noqa: F401  (same layer stack, prediction_net.py:6-26)


class Prediction_Decoder(_ArenaModule):
    def __init__(self, input_size, hidden_size, num_layers, output_size, pred_length, dropout=0.5,
                 teacher_forcing_ratio=0.5):
        super().__init__()
        self.pred_length = pred_length
        self.teacher_forcing_ratio = teacher_forcing_ratio
        self.hidden_size = hidden_size
        self.decoder = DecoderRNN(input_size, hidden_size, output_size, num_layers, dropout)
