"""Surface of reference afldm/af_libs/torch_utils (only `ops.upfirdn2d`, the one operator the AF-LDM
shift harness reaches; persistence / training_stats / custom_ops are StyleGAN3 training plumbing)."""
