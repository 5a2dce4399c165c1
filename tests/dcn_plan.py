"""Which DCN backward calls CAN take the one-launch data-gradient walk (`dcn_bwd_data_kernel`): an independent
restatement of the window rule in csrc/dcn.hip make_plan, so that a test that forces the one-launch form
(`hip_runtime.dcn_fused_min_tiles(1)`) can assert that the kernel it means to check really ran -- and that the cases
that fall back (width 1, windows beyond the LDS budget) are the expected ones."""


def one_launch_possible(H, W, k=3, s=1, p=1, d=1, dg=1):
    # (deformable_group > 1 is composed of deformable_group = 1 calls since round 6: the same rule per group)
    if W < 2:
        return False
    Wo = (W + 2 * p - (d * (k - 1) + 1)) // s + 1
    tc = 64
    while tc > 16 and tc // 2 >= Wo:
        tc //= 2
    if tc == 64 and (Wo + 31) // 32 * 32 < (Wo + 63) // 64 * 64:      # (round 6: the tile width that pads the row less)
        tc = 32
    tr = 256 // tc
    margin = 2
    wr = (tr - 1) * s + (k - 1) * d + 2 * margin + 1
    wc = (tc - 1) * s + (k - 1) * d + 2 * margin + 1
    claim = ((wr + 1) * (wc + 1) + 15) // 16 * 16
    return 16 * wr * wc * 4 + 4096 + 4 * claim <= 53 * 1024


def backward_walk_kernels(names):
    """-> 'one_launch' | 'two_kernels' | 'generic' from the kernel names of a launch log (hip_runtime.launch_log)."""
    joined = ' '.join(names)
    if 'dcn_bwd_data_kernel' in joined:
        assert 'dcn_col2im_kernel' not in joined and 'dcn_coord_grad_kernel' not in joined, names
        assert 'dcn_prep_kernel' in joined, names
        return 'one_launch'
    if 'dcn_col2im_kernel' in joined and 'dcn_coord_grad_kernel' in joined:
        return 'two_kernels'
    assert 'dcn_naive_bwd_kernel' in joined, names
    return 'generic'
