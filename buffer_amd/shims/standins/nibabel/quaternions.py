from buffer_amd.evaluate import mat2quat  # noqa: F401  (the restatement the RR evaluator uses)
