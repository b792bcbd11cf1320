"""Placeholder import surface; the full trainer is defined below in this module."""
from b2t_train_step import TrainStep, GradReducer, cosine_lr_factor, param_group_of, bucket_spans  # noqa: F401
