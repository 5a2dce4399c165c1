"""Kernel -> the oracle-value tests that select it.

Every kernel of libcenternet_uda_hip.so that one benched training step launches (tests/test_zz_kernel_coverage.py runs
that step under the library's launch log) must have an entry here naming tests that compare ITS results with the
oracle / the reference's golden vectors / an fp64 or CPU-torch restatement -- and the coverage test checks, from the
launch log recorded while those very tests ran (tests/conftest.py), that they did launch it.  A new kernel on the hot
path without a value test fails the suite; so does a size rule that moves an existing test off the kernel it claims
(round 3: the one-launch DCN data-gradient walk ran 11 of the 16 DCN backwards of the bench and no oracle test).

Keys: the kernel's symbol name with the namespace prefix and the argument list dropped (what
`tests/test_zz_kernel_coverage.py::short` makes of the demangled name).  Values: pytest node-id prefixes.
"""

MANIFEST = {
    'dcnw_fwd_kernel<64, 32>': [
        'tests/test_gpu_fullsize.py::test_full_size_dcn_layer_matches_the_oracle[64to64_128sq',
        'tests/test_gpu_fullsize.py::test_full_size_dcn_layer_matches_the_oracle[128to64_64sq',
    ],
    'dcnw_fwd_kernel<64, 16>': [
        'tests/test_gpu_dcn.py::test_forward_backward_vs_oracle[dla_64',
    ],
    'hwgrad_s2_kernel': [
        'tests/test_gpu_ops.py::test_conv2d_fwd_bwd[l1_s2_',
        'tests/test_gpu_ops.py::test_conv2d_fwd_bwd[l2_s2_',
    ],
    'hwgrad_kernel<128, false>': [
        'tests/test_gpu_ops.py::test_conv2d_fwd_bwd[offset27_w128',
    ],
    'hwgrad_kernel<16, false>': [
        'tests/test_gpu_ops.py::test_conv2d_fwd_bwd[offset18_w16',
    ],
    'hwgrad_kernel<32, false>': [
        'tests/test_gpu_ops.py::test_conv2d_fwd_bwd[offset27_w32',
    ],
    'hwgrad_kernel<64, false>': [
        'tests/test_gpu_ops.py::test_conv2d_fwd_bwd[offset27_w64',
    ],
    'hwgrad_kernel<32, true>': [
        'tests/test_gpu_ops.py::test_conv2d_fwd_bwd[offset27_w160',
        'tests/test_gpu_ops.py::test_conv2d_fwd_bwd[offset27_w96',
    ],
    'hwgrad_kernel<16, true>': [
        'tests/test_gpu_ops.py::test_conv2d_fwd_bwd[offset27_w80',
    ],
    'hwgrad_kernel<8, true>': [
        'tests/test_gpu_ops.py::test_conv2d_fwd_bwd[offset27_w40',
    ],
    # round 6: split-K of the starved long-K forward-type GEMMs, and the parity classes of a strided input gradient in one launch
    'igemm_fwd_splitk_kernel<32, ConvFwdBufLoader>': [
        'tests/test_gpu_ops.py::test_conv2d_fwd_bwd[sk_512to27_16sq',
        'tests/test_gpu_ops.py::test_conv2d_fwd_bwd[disc_last',
    ],
    'igemm_fwd_ws_splitk_kernel<64, ConvFwdBufLoader, 16>': [
        'tests/test_gpu_ops.py::test_conv2d_fwd_bwd[sk_256to64_8sq',
    ],
    'igemm_fwd_ws_splitk_kernel<128, ConvFwdBufLoader, 16>': [
        'tests/test_gpu_ops.py::test_conv2d_fwd_bwd[sk_disc_',
    ],
    'igemm_fwd_ws_splitk_kernel<64, ConvDgradBufLoader, 16>': [
        'tests/test_gpu_ops.py::test_conv2d_fwd_bwd[head_relu',
    ],
    'igemm_fwd_ws_splitk_kernel<128, ConvDgradBufLoader, 16>': [
        'tests/test_gpu_ops.py::test_conv2d_fwd_bwd[sk_256to64_8sq',
    ],
    'igemm_fwd_ws_splitk_kernel<64, ConvDgradClassBufLoader, 16>': [
        'tests/test_gpu_dla.py::test_base_step_dla_configs1_plain_1e4',
        'tests/test_gpu_dla.py::test_uda_step128_plain_1e4',
    ],
    'igemm_fwd_ws_splitk_kernel<128, ConvDgradClassBufLoader, 16>': [
        'tests/test_gpu_ops.py::test_conv2d_fwd_bwd[sk_disc_',
        'tests/test_gpu_ops.py::test_conv2d_fwd_bwd[c512',
    ],
    'splitk_reduce_kernel<ConvFwdBufLoader>': [
        'tests/test_gpu_ops.py::test_conv2d_fwd_bwd[sk_',
    ],
    'splitk_reduce_kernel<ConvDgradBufLoader>': [
        'tests/test_gpu_ops.py::test_conv2d_fwd_bwd[sk_256to64_8sq',
        'tests/test_gpu_ops.py::test_conv2d_fwd_bwd[head_relu',
    ],
    'splitk_reduce_kernel<ConvDgradClassBufLoader>': [
        'tests/test_gpu_ops.py::test_conv2d_fwd_bwd[sk_disc_',
    ],
    'igemm_fwd_kernel<32, ConvDgradClassBufLoader, false>': [      # (round 6: only where another class of the same call cuts K)
        'tests/test_gpu_ops.py::test_conv2d_fwd_bwd[sk_s2_mixed',
        'tests/test_gpu_dla.py::test_base_step_dla_configs1_plain_1e4',
    ],
    'igemm_fwd_classes_kernel<32>': [
        'tests/test_gpu_ops.py::test_conv2d_fwd_bwd[s2_even',
        'tests/test_gpu_ops.py::test_conv2d_fwd_bwd[disc4x4',
    ],
    'igemm_fwd_ws_classes_kernel<64>': [
        'tests/test_gpu_fullsize.py::test_full_size_strided_convolution_matches_fp64[64to128_128sq',
    ],
    'igemm_fwd_ws_classes_kernel<128>': [
        'tests/test_gpu_fullsize.py::test_full_size_strided_convolution_matches_fp64[128to256_64sq',
    ],
    'hconv_kernel<32, 256, HconvFwd>': [
        'tests/test_gpu_fullsize.py::test_full_size_halo_tile_convolutions_match_fp64[64to27_128sq',
        'tests/test_gpu_fullsize.py::test_full_size_halo_tile_convolutions_match_fp64[128to27_64sq',
    ],
    'hconv_kernel<32, 128, HconvFwd>': [
        'tests/test_gpu_ops.py::test_halo_tile_convolution_3x3',
    ],
    'add_kernel': [           # gradient fan-in sums that no epilogue takes over (hip_runtime.fanout)
        'tests/test_gpu_fanout.py::test_forked_graph_has_the_gradients_of_the_plain_graph[unused_alias',
        'tests/test_gpu_fanout.py::test_forked_graph_has_the_gradients_of_the_plain_graph[foreign_consumer',
    ],
    'adam_kernel': [
        'tests/test_gpu_ops.py::test_adam_matches_torch_and_skips_untouched_params',
        'tests/test_gpu_dla.py::test_uda_step128_plain_1e4',
        'tests/test_gpu_dla.py::test_base_step_dla_configs1_plain_1e4',
    ],
    'bn_apply_kernel': [
        'tests/test_gpu_ops.py::test_batch_norm_backward_gate_recomputed_from_x_is_the_gate_read_from_y',
        'tests/test_gpu_ops.py::test_batch_norm_train_fwd_bwd_and_running_stats',
        'tests/test_gpu_fuzz.py::test_batch_norm_random_geometry',
    ],
    'bn_bwd_apply_kernel': [
        'tests/test_gpu_ops.py::test_batch_norm_backward_gate_recomputed_from_x_is_the_gate_read_from_y',
        'tests/test_gpu_ops.py::test_batch_norm_train_fwd_bwd_and_running_stats',
        'tests/test_gpu_fuzz.py::test_batch_norm_random_geometry',
    ],
    'bn_reduce_kernel<0>': [
        'tests/test_gpu_ops.py::test_batch_norm_backward_gate_recomputed_from_x_is_the_gate_read_from_y',
        'tests/test_gpu_ops.py::test_batch_norm_train_fwd_bwd_and_running_stats',
        'tests/test_gpu_fuzz.py::test_batch_norm_random_geometry',
    ],
    'bn_reduce_kernel<1>': [
        'tests/test_gpu_ops.py::test_batch_norm_backward_gate_recomputed_from_x_is_the_gate_read_from_y',
        'tests/test_gpu_ops.py::test_batch_norm_train_fwd_bwd_and_running_stats',
        'tests/test_gpu_fuzz.py::test_batch_norm_random_geometry',
    ],
    'conv1x1_dgrad_act_kernel<2>': [
        'tests/test_gpu_ops.py::test_head_pair_is_one_tape_node_with_the_two_layers_values',
        'tests/test_gpu_dla.py::test_uda_step128_plain_1e4',
        'tests/test_gpu_dla.py::test_base_step_dla_configs1_plain_1e4',
    ],
    'conv1x1_dgrad_act_kernel<6>': [
        'tests/test_gpu_ops.py::test_head_pair_is_one_tape_node_with_the_two_layers_values',
        'tests/test_gpu_dla.py::test_uda_step128_plain_1e4',
        'tests/test_gpu_dla.py::test_base_step_dla_configs1_plain_1e4',
    ],
    'copy_channels_kernel': [
        'tests/test_gpu_ops.py::test_cat_add_split',
        'tests/test_gpu_dla.py::test_uda_step128_plain_1e4',
        'tests/test_gpu_dla.py::test_base_step_dla_configs1_plain_1e4',
    ],
    'dcn_bwd_data_kernel<true>': [
        'tests/test_gpu_fullsize.py::test_full_size_dcn_layer_matches_the_oracle[64to64_128sq_one_launch',
        'tests/test_gpu_fullsize.py::test_full_size_dcn_layer_matches_the_oracle[128to64_64sq_one_launch',
        'tests/test_gpu_dcn.py::test_forward_backward_vs_oracle',
        'tests/test_gpu_fuzz.py::test_dcn_random_geometry_vs_oracle',
    ],
    'dcn_bwd_data_kernel<false>': [      # (the plain column-gradient layout: C % 4 != 0 or an output-channel count the quad GEMM does not take)
        'tests/test_gpu_dcn.py::test_forward_backward_vs_oracle',
        'tests/test_gpu_fuzz.py::test_dcn_random_geometry_vs_oracle',
        'tests/test_gpu_dcn.py::test_known_answer_validity_window_open_at_minus_one_and_H',
    ],
    'dcn_col2im_kernel<true>': [
        'tests/test_gpu_fullsize.py::test_full_size_dcn_layer_matches_the_oracle',
        'tests/test_gpu_dcn.py::test_forward_backward_vs_oracle',
    ],
    'dcn_col2im_kernel<false>': [      # (the plain column-gradient layout: C % 4 != 0 or an output-channel count the quad GEMM does not take)
        'tests/test_gpu_dcn.py::test_forward_backward_vs_oracle',
        'tests/test_gpu_fuzz.py::test_dcn_random_geometry_vs_oracle',
        'tests/test_gpu_dcn.py::test_autograd_module_path_and_argument_order',
    ],
    'dcn_coord_grad_kernel<true>': [
        'tests/test_gpu_fullsize.py::test_full_size_dcn_layer_matches_the_oracle',
        'tests/test_gpu_dcn.py::test_forward_backward_vs_oracle',
    ],
    'dcn_coord_grad_kernel<false>': [      # (the plain column-gradient layout: C % 4 != 0 or an output-channel count the quad GEMM does not take)
        'tests/test_gpu_dcn.py::test_forward_backward_vs_oracle',
        'tests/test_gpu_fuzz.py::test_dcn_random_geometry_vs_oracle',
        'tests/test_gpu_dcn.py::test_autograd_module_path_and_argument_order',
    ],
    'dcn_prep_kernel': [
        'tests/test_gpu_fullsize.py::test_full_size_dcn_layer_matches_the_oracle',
        'tests/test_gpu_dcn.py::test_autograd_module_path_and_argument_order',
        'tests/test_gpu_dcn.py::test_forward_backward_vs_oracle',
    ],
    'dcn_sample_kernel': [
        'tests/test_gpu_dcn.py::test_forward_backward_vs_oracle',
        'tests/test_gpu_dcn.py::test_known_answer_half_pixel_offsets_are_box_blurs',
        'tests/test_gpu_dcn.py::test_known_answer_linear_ramp_gradients',
    ],
    'dwconvt_bwd_k4s2_kernel': [
        'tests/test_gpu_ops.py::test_depthwise_conv_transpose',
        'tests/test_gpu_ops.py::test_depthwise_conv_transpose_with_summand_is_the_separate_add',
        'tests/test_gpu_dla.py::test_uda_step128_plain_1e4',
    ],
    'dwconvt_bwd_kernel<8>': [
        'tests/test_gpu_ops.py::test_depthwise_conv_transpose',
        'tests/test_gpu_ops.py::test_depthwise_conv_transpose_with_summand_is_the_separate_add',
        'tests/test_gpu_dla.py::test_uda_step128_plain_1e4',
    ],
    'dwconvt_fwd_f_kernel<2, 1>': [
        'tests/test_gpu_ops.py::test_depthwise_conv_transpose',
        'tests/test_gpu_ops.py::test_depthwise_conv_transpose_with_summand_is_the_separate_add',
        'tests/test_gpu_dla.py::test_uda_step128_plain_1e4',
    ],
    'dwconvt_fwd_f_kernel<4, 4>': [
        'tests/test_gpu_ops.py::test_depthwise_conv_transpose',
        'tests/test_gpu_ops.py::test_depthwise_conv_transpose_with_summand_is_the_separate_add',
    ],
    'dwconvt_fwd_k4s2_kernel': [
        'tests/test_gpu_ops.py::test_depthwise_conv_transpose',
        'tests/test_gpu_ops.py::test_depthwise_conv_transpose_with_summand_is_the_separate_add',
    ],
    'dwconvt_wsum_kernel': [
        'tests/test_gpu_ops.py::test_depthwise_conv_transpose',
        'tests/test_gpu_ops.py::test_depthwise_conv_transpose_with_summand_is_the_separate_add',
        'tests/test_gpu_dla.py::test_uda_step128_plain_1e4',
    ],
    'focal_bwd_kernel': [
        'tests/test_gpu_losses.py::test_detection_loss_golden',
        'tests/test_gpu_losses.py::test_full_size_losses_vs_oracle_cfg3',
        'tests/test_gpu_losses.py::test_keypoint_detection_loss_golden',
    ],
    'focal_finalize_kernel': [
        'tests/test_gpu_losses.py::test_detection_loss_golden',
        'tests/test_gpu_losses.py::test_full_size_losses_vs_oracle_cfg3',
        'tests/test_gpu_losses.py::test_keypoint_detection_loss_golden',
    ],
    'focal_fwd_kernel': [
        'tests/test_gpu_losses.py::test_detection_loss_golden',
        'tests/test_gpu_losses.py::test_full_size_losses_vs_oracle_cfg3',
        'tests/test_gpu_losses.py::test_keypoint_detection_loss_golden',
    ],
    'igemm_fwd_ws_kernel<128, DcnColsBufLoader, 16>': [
        'tests/test_gpu_fullsize.py::test_full_size_dcn_layer_matches_the_oracle[256to256_32sq',
    ],
    'igemm_fwd_kernel<128, DcnFwdLoaderT<true>, false>': [
        'tests/test_gpu_fullsize.py::test_full_size_dcn_layer_matches_the_oracle[128to128_64sq',
    ],
    'igemm_fwd_kernel<32, ConvDgradBufLoader, false>': [
        'tests/test_gpu_fullsize.py::test_full_size_1x1_convolutions_match_fp64_and_repeat',
        'tests/test_gpu_ops.py::test_conv2d_fwd_bwd',
        'tests/test_gpu_ops.py::test_head_pair_is_one_tape_node_with_the_two_layers_values',
    ],
    'igemm_fwd_kernel<32, ConvFwdBufLoader, false>': [
        'tests/test_gpu_fullsize.py::test_full_size_1x1_convolutions_match_fp64_and_repeat',
        'tests/test_gpu_ops.py::test_conv2d_fwd_bwd',
        'tests/test_gpu_ops.py::test_head_pair_is_one_tape_node_with_the_two_layers_values',
    ],
    'igemm_fwd_kernel<32, DcnColsBufLoader, false>': [
        'tests/test_gpu_dcn.py::test_forward_backward_vs_oracle',
        'tests/test_gpu_dcn.py::test_known_answer_half_pixel_offsets_are_box_blurs',
        'tests/test_gpu_dcn.py::test_known_answer_linear_ramp_gradients',
    ],
    'igemm_fwd_ws_kernel<64, DcnColsBufLoader, 16>': [
        'tests/test_gpu_fullsize.py::test_full_size_dcn_layer_matches_the_oracle[256to128_32sq',
    ],
    'igemm_fwd_kernel<64, DcnFwdLoaderT<true>, false>': [      # (since round 4: only layers the window kernel does not take)
        'tests/test_gpu_fullsize.py::test_full_size_dcn_layer_matches_the_oracle[32to64_100sq',
    ],
    'igemm_fwd_shortk_kernel<128, ConvFwdBufLoader, 64>': [
        'tests/test_gpu_fullsize.py::test_full_size_1x1_convolutions_match_fp64_and_repeat',
        'tests/test_gpu_ops.py::test_conv2d_forward_with_quad_interleaved_output_rows[shortk',
    ],
    # round 6: DLA's Root as a 1x1 convolution over the concatenation without the concatenation (cnuda_conv2d_cat_*)
    'igemm_fwd_ws_kernel<128, ConvFwdCatLoader, 16>': [
        'tests/test_gpu_ops.py::test_conv1x1_over_a_concatenation_without_the_concatenation',
    ],
    'igemm_fwd_ws_kernel<64, ConvFwdCatLoader, 16>': [
        'tests/test_gpu_ops.py::test_conv1x1_over_a_concatenation_without_the_concatenation',
    ],
    'igemm_fwd_ws_kernel<128, ConvDgradCatLoader, 16>': [
        'tests/test_gpu_ops.py::test_conv1x1_over_a_concatenation_without_the_concatenation',
    ],
    'igemm_fwd_ws_kernel<64, ConvDgradCatLoader, 16>': [
        'tests/test_gpu_ops.py::test_conv1x1_over_a_concatenation_without_the_concatenation',
    ],
    'igemm_wgrad_ws_kernel<ConvWCatLoader, 128, 128>': [
        'tests/test_gpu_ops.py::test_conv1x1_over_a_concatenation_without_the_concatenation',
    ],
    'igemm_wgrad_ws_kernel<ConvWCatLoader, 128, 64>': [
        'tests/test_gpu_ops.py::test_conv1x1_over_a_concatenation_without_the_concatenation',
    ],
    'igemm_wgrad_ws_kernel<ConvWCatLoader, 64, 128>': [
        'tests/test_gpu_ops.py::test_conv1x1_over_a_concatenation_without_the_concatenation',
    ],
    # round 6: the DCN column gradient with its rows interleaved in quads (cnuda_conv2d_forward_rowquads)
    'igemm_fwd_shortk_kernel<128, ConvFwdBufQuadLoader, 64>': [
        'tests/test_gpu_ops.py::test_conv2d_forward_with_quad_interleaved_output_rows[shortk',
        'tests/test_gpu_fullsize.py::test_full_size_dcn_layer_matches_the_oracle[64to64_128sq',
    ],
    'igemm_fwd_ws_kernel<128, ConvFwdBufQuadLoader, 16>': [
        'tests/test_gpu_ops.py::test_conv2d_forward_with_quad_interleaved_output_rows[ws128',
        'tests/test_gpu_fullsize.py::test_full_size_dcn_layer_matches_the_oracle',
    ],
    'igemm_fwd_ws_kernel<64, ConvFwdBufQuadLoader, 16>': [
        'tests/test_gpu_ops.py::test_conv2d_forward_with_quad_interleaved_output_rows[ws64',
    ],
    'igemm_fwd_ws_kernel<128, ConvDgradBufLoader, 16>': [
        'tests/test_gpu_fullsize.py::test_full_size_1x1_convolutions_match_fp64_and_repeat',
        'tests/test_gpu_fullsize.py::test_full_size_dcn_layer_matches_the_oracle',
    ],
    'igemm_fwd_ws_kernel<128, ConvFwdBufLoader, 16>': [
        'tests/test_gpu_fullsize.py::test_full_size_3x3_convolution_matches_fp64',
        'tests/test_gpu_fullsize.py::test_full_size_dcn_layer_matches_the_oracle',
    ],
    'igemm_fwd_ws_kernel<64, ConvDgradBufLoader, 16>': [
        'tests/test_gpu_fullsize.py::test_full_size_1x1_convolutions_match_fp64_and_repeat',
        'tests/test_gpu_fullsize.py::test_full_size_3x3_convolution_matches_fp64',
        'tests/test_gpu_resnet.py::test_model_step_resnet18_config0',
    ],
    'igemm_fwd_ws_kernel<64, ConvFwdBufLoader, 16>': [
        'tests/test_gpu_fullsize.py::test_full_size_1x1_convolutions_match_fp64_and_repeat',
        'tests/test_gpu_dla.py::test_uda_step128_plain_1e4',
    ],
    'igemm_wgrad_kernel<ConvWBufLoader, 32, 128>': [
        'tests/test_gpu_fullsize.py::test_full_size_1x1_convolutions_match_fp64_and_repeat',
        'tests/test_gpu_ops.py::test_conv2d_fwd_bwd',
        'tests/test_gpu_ops.py::test_head_pair_is_one_tape_node_with_the_two_layers_values',
    ],
    # (channel counts that are no multiple of 8: the pointer-arithmetic loader; multiples of 8 moved to ConvWBufLoaderC8)
    'igemm_wgrad_kernel<ConvWLoader<0>, 32, 128>': [
        'tests/test_gpu_fuzz.py::test_conv2d_random_geometry',
    ],
    'igemm_wgrad_kernel<ConvWLoader<0>, 64, 64>': [
        'tests/test_gpu_ops.py::test_conv2d_fwd_bwd[disc4x4_full',
        'tests/test_gpu_fuzz.py::test_conv2d_random_geometry',
    ],
    'igemm_wgrad_ws_kernel<DcnColWBufLoader, 64, 128>': [
        'tests/test_gpu_fullsize.py::test_full_size_dcn_layer_matches_the_oracle',
        'tests/test_gpu_dla.py::test_uda_step128_plain_1e4',
        'tests/test_gpu_dla.py::test_base_step_dla_configs1_plain_1e4',
    ],
    'igemm_wgrad_ws_kernel<DcnColWBufLoader, 64, 64>': [
        'tests/test_gpu_fullsize.py::test_full_size_dcn_layer_matches_the_oracle',
        'tests/test_gpu_dcn.py::test_autograd_module_path_and_argument_order',
        'tests/test_gpu_dla.py::test_uda_step128_plain_1e4',
    ],
    'igemm_wgrad_ws_kernel<ConvWBufLoader, 128, 128>': [
        'tests/test_gpu_fullsize.py::test_full_size_dcn_layer_matches_the_oracle',
        'tests/test_gpu_ops.py::test_conv2d_fwd_bwd',
        'tests/test_gpu_dla.py::test_uda_step128_plain_1e4',
    ],
    'igemm_wgrad_ws_kernel<ConvWBufLoader, 128, 64>': [
        'tests/test_gpu_fullsize.py::test_full_size_3x3_convolution_matches_fp64',
        'tests/test_gpu_ops.py::test_conv2d_fwd_bwd',
        'tests/test_gpu_ops.py::test_head_pair_is_one_tape_node_with_the_two_layers_values',
    ],
    'igemm_wgrad_ws_kernel<ConvWBufLoader, 64, 128>': [
        'tests/test_gpu_fullsize.py::test_full_size_1x1_convolutions_match_fp64_and_repeat',
        'tests/test_gpu_fuzz.py::test_conv_transpose2d_random_geometry',
    ],
    'igemm_wgrad_ws_kernel<ConvWBufLoader, 64, 64>': [
        'tests/test_gpu_fullsize.py::test_full_size_1x1_convolutions_match_fp64_and_repeat',
        'tests/test_gpu_ops.py::test_conv2d_fwd_bwd',
        'tests/test_gpu_fuzz.py::test_conv2d_random_geometry',
    ],
    'maxpool_bwd_kernel': [
        'tests/test_gpu_ops.py::test_maxpool',
    ],
    'maxpool_fwd_kernel': [
        'tests/test_gpu_ops.py::test_maxpool',
    ],
    'maxpool2_bwd_vec_kernel': [       # round 6: the 2 x 2 pools with 16-byte accesses
        'tests/test_gpu_ops.py::test_maxpool',
        'tests/test_gpu_dla.py::test_uda_step128_plain_1e4',
        'tests/test_gpu_dla.py::test_base_step_dla_configs1_plain_1e4',
    ],
    'maxpool2_fwd_vec_kernel': [
        'tests/test_gpu_ops.py::test_maxpool',
        'tests/test_gpu_dla.py::test_uda_step128_plain_1e4',
        'tests/test_gpu_dla.py::test_base_step_dla_configs1_plain_1e4',
    ],
    'pack_kernel': [
        'tests/test_gpu_fullsize.py::test_full_size_1x1_convolutions_match_fp64_and_repeat',
        'tests/test_gpu_fullsize.py::test_full_size_3x3_convolution_matches_fp64',
        'tests/test_gpu_fullsize.py::test_full_size_dcn_layer_matches_the_oracle',
    ],
    'pack_multi_kernel': [
        'tests/test_gpu_ops.py::test_pack_refresh_after_the_fused_adam_step',
    ],
    'pack_taps_kernel': [
        'tests/test_gpu_ops.py::test_conv2d_fwd_bwd',
        'tests/test_gpu_fuzz.py::test_conv2d_random_geometry',
        'tests/test_gpu_fuzz.py::test_conv_transpose2d_random_geometry',
    ],
    'regl1_bwd_kernel': [
        'tests/test_gpu_losses.py::test_detection_loss_golden',
        'tests/test_gpu_losses.py::test_full_size_losses_vs_oracle_cfg3',
        'tests/test_gpu_losses.py::test_keypoint_detection_loss_golden',
    ],
    'regl1_fwd_kernel': [
        'tests/test_gpu_losses.py::test_detection_loss_golden',
        'tests/test_gpu_losses.py::test_full_size_losses_vs_oracle_cfg3',
        'tests/test_gpu_losses.py::test_keypoint_detection_loss_golden',
    ],
    'scalar_finalize_kernel': [
        'tests/test_gpu_losses.py::test_full_size_losses_vs_oracle_cfg3',
        'tests/test_gpu_losses.py::test_uda_losses_golden',
        'tests/test_gpu_dla.py::test_uda_step128_plain_1e4',
    ],
    'slab_reduce_few_kernel': [
        'tests/test_gpu_ops.py::test_conv2d_fwd_bwd',
        'tests/test_gpu_dcn.py::test_forward_backward_vs_oracle',
        'tests/test_gpu_dla.py::test_uda_step128_plain_1e4',
    ],
    'slab_reduce_kernel': [
        'tests/test_gpu_fullsize.py::test_full_size_1x1_convolutions_match_fp64_and_repeat',
        'tests/test_gpu_fullsize.py::test_full_size_3x3_convolution_matches_fp64',
        'tests/test_gpu_fullsize.py::test_full_size_dcn_layer_matches_the_oracle',
    ],
    'smallc_fwd_kernel<1, 3, true>': [      # (BatchNorm + ReLU of the input applied while staging: apply on load)
        'tests/test_gpu_ops.py::test_batchnorm_applied_on_load_equals_the_materialised_activation',
    ],
    'smallc_wgrad_kernel<1, 3, 9, true>': [
        'tests/test_gpu_ops.py::test_batchnorm_applied_on_load_equals_the_materialised_activation',
    ],
    'smallc_fwd_kernel<1, 3, false>': [
        'tests/test_gpu_ops.py::test_conv2d_fwd_bwd[c16_3x3',
        'tests/test_gpu_ops.py::test_convolution_epilogue_leaves_batchnorm_statistics[B2C16H20W64Co16k3s1g1',
        'tests/test_gpu_fuzz.py::test_conv2d_random_geometry[0-B3C3H8W11Co16k3s1p1ba',
    ],
    'smallc_fwd_kernel<1, 7, false>': [
        'tests/test_gpu_ops.py::test_conv2d_fwd_bwd[stem7x7',
        'tests/test_gpu_ops.py::test_convolution_epilogue_leaves_batchnorm_statistics[B2C3H24W72Co16k7s1g2',
        'tests/test_gpu_fuzz.py::test_conv2d_random_geometry[0-B1C1H18W17Co2k7s1p3',
    ],
    'smallc_koff_kernel': [
        'tests/test_gpu_ops.py::test_conv2d_fwd_bwd',
        'tests/test_gpu_ops.py::test_head_pair_is_one_tape_node_with_the_two_layers_values',
        'tests/test_gpu_fuzz.py::test_conv2d_random_geometry',
    ],
    'smallc_pack_kernel': [
        'tests/test_gpu_ops.py::test_conv2d_fwd_bwd',
        'tests/test_gpu_ops.py::test_head_pair_is_one_tape_node_with_the_two_layers_values',
        'tests/test_gpu_fuzz.py::test_conv2d_random_geometry',
    ],
    'smallc_slab_reduce1_kernel': [
        'tests/test_gpu_ops.py::test_conv2d_fwd_bwd',
        'tests/test_gpu_ops.py::test_head_pair_is_one_tape_node_with_the_two_layers_values',
        'tests/test_gpu_fuzz.py::test_conv2d_random_geometry',
    ],
    'smallc_slab_reduce2_kernel': [
        'tests/test_gpu_ops.py::test_conv2d_fwd_bwd',
        'tests/test_gpu_ops.py::test_head_pair_is_one_tape_node_with_the_two_layers_values',
        'tests/test_gpu_fuzz.py::test_conv2d_random_geometry',
    ],
    'smallc_wgrad_kernel<1, 3, 9, false>': [
        'tests/test_gpu_ops.py::test_conv2d_fwd_bwd[c16_3x3',
        'tests/test_gpu_ops.py::test_convolution_epilogue_leaves_batchnorm_statistics[B2C16H20W64Co16k3s1g1',
    ],
    'smallc_wgrad_kernel<1, 7, 10, false>': [
        'tests/test_gpu_ops.py::test_conv2d_fwd_bwd[stem7x7',
        'tests/test_gpu_ops.py::test_convolution_epilogue_leaves_batchnorm_statistics[B2C3H24W72Co16k7s1g2',
    ],
    'softmax_loss_bwd_kernel': [
        'tests/test_gpu_losses.py::test_full_size_losses_vs_oracle_cfg3',
        'tests/test_gpu_losses.py::test_uda_losses_golden',
        'tests/test_gpu_dla.py::test_uda_step128_plain_1e4',
    ],
    'softmax_loss_fwd_kernel': [
        'tests/test_gpu_losses.py::test_full_size_losses_vs_oracle_cfg3',
        'tests/test_gpu_losses.py::test_uda_losses_golden',
        'tests/test_gpu_dla.py::test_uda_step128_plain_1e4',
    ],
    # (round 6: DCN.forward reads offsets / mask out of the offset convolution's output; the split pair remains for
    # DCNv2.forward(input, offset, mask)'s callers and CNUDA_DCN_OM=0)
    'split_offset_mask_bwd_kernel': [
        'tests/test_gpu_ops.py::test_cat_add_split',
        'tests/test_gpu_dcn.py::test_offsets_and_mask_read_out_of_the_offset_convolutions_output',
    ],
    'split_offset_mask_kernel': [
        'tests/test_gpu_ops.py::test_cat_add_split',
        'tests/test_gpu_dcn.py::test_offsets_and_mask_read_out_of_the_offset_convolutions_output',
    ],
    # round 6: the offset convolution with the mask's sigmoid in its epilogue (cnuda_conv2d_forward_rowsig), bit-identical to
    # convolution + split kernel in these tests; values against the reference fixtures through every DLA step test
    'hconv_kernel<32, 128, HconvFwdSig>': [
        'tests/test_gpu_dcn.py::test_offsets_and_mask_read_out_of_the_offset_convolutions_output[halo_tile_offsets',
    ],
    'hconv_kernel<32, 256, HconvFwdSig>': [
        'tests/test_gpu_dcn.py::test_offsets_and_mask_read_out_of_the_offset_convolutions_output[halo_tile_256_offsets',
    ],
    'igemm_fwd_splitk_kernel<32, ConvFwdBufSigLoader>': [
        'tests/test_gpu_dcn.py::test_offsets_and_mask_read_out_of_the_offset_convolutions_output[split_k',
    ],
    'splitk_reduce_kernel<ConvFwdBufSigLoader>': [
        'tests/test_gpu_dcn.py::test_offsets_and_mask_read_out_of_the_offset_convolutions_output[split_k',
    ],
    'igemm_fwd_kernel<32, ConvFwdBufSigLoader, false>': [
        'tests/test_gpu_dcn.py::test_offsets_and_mask_read_out_of_the_offset_convolutions_output[small_odd',
    ],
    # ---- kernels only the other BASELINE configs launch (round 5: the guard runs configs[0], [1], [3], [4] at full size too) ----
    'act_bwd_kernel': [           # LeakyReLU(0.2) gate of the discriminator's convolutions (configs[4])
        'tests/test_gpu_ops.py::test_conv2d_fwd_bwd[disc4x4',
        'tests/test_gpu_fuzz.py::test_conv2d_random_geometry',
    ],
    'bce_const_fwd_kernel': ['tests/test_gpu_losses.py::test_uda_losses_golden'],
    'bce_const_bwd_kernel': ['tests/test_gpu_losses.py::test_uda_losses_golden'],
    'entropy_map_fwd_kernel': ['tests/test_gpu_losses.py::test_uda_losses_golden'],
    'entropy_map_bwd_kernel': ['tests/test_gpu_losses.py::test_uda_losses_golden'],
    'conv1x1_dgrad_act_kernel<3>': [          # the rotated-box `wh` head (3 outputs)
        'tests/test_gpu_dla.py::test_dla_forward_backward_golden[rot',
        'tests/test_gpu_dla.py::test_uda_step128_plain_1e4[advent',
    ],
    'igemm_fwd_kernel<32, ConvDgradLoader, false>': [
        'tests/test_gpu_fuzz.py::test_conv2d_random_geometry',
        'tests/test_gpu_fanout.py::test_backward_data_add_with_the_output_as_addend',
    ],
    'igemm_fwd_ws_kernel<64, ConvFwdLoader<false>, 16>': [
        'tests/test_gpu_ops.py::test_conv2d_fwd_bwd[disc4x4_full',
        'tests/test_gpu_ops.py::test_conv2d_fwd_bwd[resnet_stem_full',
    ],
    'maxpool_win_fwd_kernel': ['tests/test_gpu_resnet.py::test_max_pool_window_matches_torch'],
    'maxpool_win_bwd_kernel': ['tests/test_gpu_resnet.py::test_max_pool_window_matches_torch'],
    'bn_fold_stats_kernel': [      # BatchNorm statistics from the producing GEMM's epilogue (round 5)
        'tests/test_gpu_ops.py::test_convolution_epilogue_leaves_batchnorm_statistics',
    ],
    'igemm_wgrad_kernel<ConvWBufLoaderC8, 32, 128>': [       # 8 | C, 64 does not: the 16 -> 32 stride-2 convolution (round 5)
        'tests/test_gpu_ops.py::test_conv2d_fwd_bwd[c16_s2',
        'tests/test_gpu_ops.py::test_convolution_epilogue_leaves_batchnorm_statistics',
    ],
    'igemm_wgrad_kernel<ConvWBufLoaderC8, 64, 64>': [        # the 32 -> 64 stride-2 convolution
        'tests/test_gpu_ops.py::test_conv2d_fwd_bwd[s2_even',
        'tests/test_gpu_ops.py::test_convolution_epilogue_leaves_batchnorm_statistics',
    ],
    'dgrad_s2_c16_kernel': [          # input gradient of the 16 -> 32 stride-2 convolution (level1), round 5
        'tests/test_gpu_ops.py::test_conv2d_fwd_bwd[c16_s2',
        'tests/test_gpu_ops.py::test_convolution_epilogue_leaves_batchnorm_statistics',
        'tests/test_gpu_ops.py::test_stride2_sixteen_channel_input_gradient',
    ],
    'dgrad_s2_c16_pack_kernel': [
        'tests/test_gpu_ops.py::test_conv2d_fwd_bwd[c16_s2',
        'tests/test_gpu_ops.py::test_stride2_sixteen_channel_input_gradient',
    ],
    # the forward GEMMs that leave BatchNorm statistics with their output (their own loader type, round 5)
    'igemm_fwd_ws_kernel<128, ConvFwdBufStatsLoader, 16>': [
        'tests/test_gpu_ops.py::test_convolution_epilogue_leaves_batchnorm_statistics[B4C16H128W128Co128',
        'tests/test_gpu_fullsize.py::test_full_size_step_properties',
    ],
    'igemm_fwd_ws_kernel<64, ConvFwdBufStatsLoader, 16>': [
        'tests/test_gpu_ops.py::test_convolution_epilogue_leaves_batchnorm_statistics[B4C16H128W128Co64',
        'tests/test_gpu_fullsize.py::test_full_size_step_properties',
    ],
    'igemm_fwd_kernel<32, ConvFwdBufStatsLoader, false>': [
        'tests/test_gpu_ops.py::test_convolution_epilogue_leaves_batchnorm_statistics',
        'tests/test_gpu_dla.py::test_uda_step128_plain_1e4',
    ],
}
