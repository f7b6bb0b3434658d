#!/bin/bash
# Marginal cost of each kernel family INSIDE the two-stream hipGraph step: tools/ablate.py runs bench.py with the family's launches
# skipped. The arithmetic is wrong in these runs (the NaN guard may skip the optimiser update); only the step time is read.
# Usage (GPU box): tools/ablate.sh > gpurun_out/ablate.txt
run() { python tools/ablate.py "$2" -- --no-cpu-baseline --no-roofline --no-other 2>&1 >/dev/null | grep timed | sed "s/^/$1: /"; }
run "baseline" "nsid_version"
run "no bn_finalize (fwd)" "nsid_bn_finalize,nsid_bn_finalize_deferred"
run "no bn_bwd_finalize" "nsid_bn_bwd_finalize,nsid_bn_bwd_finalize_fused"
run "no bn_bwd_apply" "nsid_bn_bwd_apply"
run "no bn_apply" "nsid_bn_apply"
run "no weight gradients" "nsid_linear_bwd_weight,nsid_linear_bwd_weight_grouped,nsid_downsample3_bwd_weight"
run "no backward-data" "nsid_linear_bwd_data,nsid_linear_bwd_data_bn,nsid_linear_bwd_data_bnapply,nsid_downsample3_bwd_data"
run "no forward GEMMs" "nsid_linear_fwd,nsid_linear_fwd_res,nsid_downsample3_fwd"
run "no kNN" "nsid_knn_graph"
run "no aggregation fwd" "nsid_mr_aggregate_fwd"
run "no aggregation bwd" "nsid_mr_aggregate_bwd,nsid_mr_aggregate_bwd_bn"
run "no col_reduce (bn_bwd_reduce)" "nsid_bn_bwd_reduce"
run "no NT-Xent" "nsid_ntxent_fwd_bwd"
run "no patchify" "nsid_peak_patchify_fwd,nsid_peak_patchify_bwd,nsid_peak_patchify_bwd_ws"
run "no node mean, l2norm, elu" "nsid_node_mean_fwd,nsid_node_mean_bwd,nsid_l2norm_fwd,nsid_l2norm_bwd,nsid_elu_bwd"
run "no optimiser" "nsid_adam_step,nsid_sumsq_partial,nsid_fill_zero,nsid_f32_to_bf16,nsid_ds_prepack"
run "baseline again" "nsid_version"
