"""Defaults of xt/model/impala/default_config.py."""
LR = 0.0003
ENTROPY_LOSS = 0.01
HIDDEN_SIZE = 128
NUM_LAYERS = 1
GAMMA = 0.99
