"""hip_runtime.fanout: a tensor's consumers get one alias each and their gradients meet in a slot (a convolution's
input-gradient epilogue, a DCN's atomics, or the library's own add) instead of in autograd's accumulation passes
(torch's engine: one at::add per extra consumer).  Every graph below is built twice -- with forks and plainly, where the
engine does the sums -- and the gradients must agree to float rounding of a different summation order; the launch log
says which kernels did the sums."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda', 0) if torch.cuda.is_available() else None


def _close(a, b, tol=2e-6):
    scale = max(1.0, b.abs().max().item())
    assert (a - b).abs().max().item() <= tol * scale, ((a - b).abs().max().item(), scale)


def _leaves(*shapes, seed=0):
    g = torch.Generator().manual_seed(seed)
    return [torch.randn(*s, generator=g).to(DEV).requires_grad_(True) for s in shapes]


def _run(build, leaves, forked):
    import hip_runtime.fanout as fo
    for t in leaves:
        t.grad = None
    fork = fo.fork if forked else (lambda x, n=2: (x,) * n)
    out = build(fork, *leaves)
    torch.manual_seed(1)
    out.backward(torch.randn_like(out))
    return [None if t.grad is None else t.grad.clone() for t in leaves]


@pytest.mark.parametrize('case', ['block', 'tree_node', 'pool_and_conv', 'foreign_consumer', 'unused_alias', 'three_convs'])
def test_forked_graph_has_the_gradients_of_the_plain_graph(case):
    import hip_runtime as hr
    from hip_runtime import ops
    x, w1, w2, w3, gam, bet = _leaves((2, 32, 16, 16), (32, 32, 3, 3), (32, 32, 3, 3), (16, 64, 1, 1), (32,), (32,))
    rm, rv = torch.zeros(32, device=DEV), torch.ones(32, device=DEV)

    def bn(t, res=None):
        return ops.batch_norm_act(t, gam, bet, rm.clone(), rv.clone(), True, residual=res, relu=True)

    def build(fork, x, w1, w2, w3, gam, bet):
        h = ops.conv2d(x, w1, None, 1, 1)                    # x itself has one consumer; h is what fans out
        if case == 'block':                                  # conv + identity skip (BasicBlock)
            a, b = fork(h, 2)
            return bn(ops.conv2d(a, w2, None, 1, 1), res=b)
        if case == 'tree_node':                              # conv + skip inside a block, and a concatenation outside: nested
            nxt, root = fork(h, 2)
            a, b = fork(nxt, 2)
            y = bn(ops.conv2d(a, w2, None, 1, 1), res=b)
            return ops.conv2d(ops.cat_channels([y, root]), w3, None, 1, 0)
        if case == 'pool_and_conv':                          # max-pool + strided conv (a Tree's input)
            a, b = fork(h, 2)
            return ops.max_pool2d(a, 2) + ops.conv2d(b, w2, None, 2, 1)
        if case == 'foreign_consumer':                       # a consumer that knows nothing about slots
            a, b, c = fork(h, 3)
            return ops.conv2d(a, w2, None, 1, 1) + torch.tanh(b) * 0.5 + ops.conv2d(c, w1, None, 1, 1)
        if case == 'unused_alias':                           # one alias never reaches the loss
            a, b, c = fork(h, 3)
            _ = ops.conv2d(c, w2, None, 1, 1)
            return ops.conv2d(a, w2, None, 1, 1) + bn(b)
        a, b, c = fork(h, 3)                                 # three_convs
        return ops.conv2d(a, w2, None, 1, 1) + ops.conv2d(b, w1, None, 1, 1) + ops.conv2d(c, w2, None, 1, 1)

    leaves = [x, w1, w2, w3, gam, bet]
    with hr.launch_log() as log:
        got = _run(build, leaves, True)
    forked_kernels = dict(log.counts)
    want = _run(build, leaves, False)
    for a, b in zip(got, want):
        assert (a is None) == (b is None)
        if a is not None:
            _close(a, b)
    adds = sum(n for k, n in forked_kernels.items() if 'add_kernel' in k)
    # the sums a convolution's epilogue (or a first writer's plain store) takes over need no pass of their own
    assert adds == {'block': 0, 'tree_node': 0, 'pool_and_conv': 0, 'foreign_consumer': 1, 'unused_alias': 1,
                    'three_convs': 0}[case], forked_kernels


def test_dcn_input_meets_the_offset_convolution_in_the_slot():
    import hip_runtime as hr
    from libs.DCNv2.dcn_v2 import DCN
    torch.manual_seed(3)
    m = DCN(32, 32, kernel_size=(3, 3), stride=1, padding=1, dilation=1, deformable_groups=1).to(DEV)
    with torch.no_grad():
        m.conv_offset_mask.weight.normal_(0, 0.05)
    x = torch.randn(2, 32, 12, 20, device=DEV, requires_grad=True)
    g = torch.randn(2, 32, 12, 20, device=DEV)
    with hr.launch_log() as log:
        m(x).backward(g)
    got = x.grad.clone()
    assert not any('add_kernel' in k for k in log.names), log.names
    # the same layer with the two consumers on the plain tensor: autograd's engine sums
    from hip_runtime import ops
    from libs.DCNv2.dcn_v2 import dcn_v2_conv
    x.grad = None
    off, mask = ops.split_offset_mask(m.conv_offset_mask(x))
    dcn_v2_conv(x, off, mask, m.weight, m.bias, m.stride, m.padding, m.dilation, 1).backward(g)
    _close(got, x.grad, 1e-5)


def test_head_on_the_leading_images_leaves_a_whole_gradient():
    import hip_runtime.fanout as fo
    from hip_runtime import nn as hnn
    torch.manual_seed(5)
    heads = [hnn.Head(hnn.Conv2d(16, 32, 3, padding=1, bias=True, act_slope=0.0), hnn.Slot(), hnn.Conv2d(32, c, 1)).to(DEV)
             for c in (2, 3)]
    x = torch.randn(4, 16, 8, 8, device=DEV, requires_grad=True)
    h = x * 1.0
    for order in ((0, 1), (1, 0)):                 # whichever head's backward comes first
        x.grad = None
        a, b = fo.fork(x * 1.0, 2)
        outs = {0: lambda: heads[0](a), 1: lambda: heads[1](b, lead=2)}
        ys = [outs[i]() for i in order]
        sum(y.sum() * (i + 1.5) for i, y in zip(order, ys)).backward()
        got = x.grad.clone()
        x.grad = None
        h = x * 1.0
        (heads[0](h).sum() * 1.5 + heads[1](h[:2]).sum() * 2.5).backward()
        _close(got, x.grad, 1e-5)
        assert got[2:].abs().sum() > 0


@pytest.mark.parametrize('ch', [16, 32])
def test_block_at_sixteen_channels_accumulates_in_place(ch):
    """A forked BasicBlock whose convolutions have <= 16 channels: the slot's owned buffer is BOTH the addend and the output
    of conv1's input gradient.  The LDS-tile kernels (smallc) store before they add and must not take that call."""
    from hip_runtime import ops
    x, w0, w1, w2, gam, bet = _leaves((2, ch, 24, 24), (ch, ch, 3, 3), (ch, ch, 3, 3), (ch, ch, 3, 3), (ch,), (ch,))
    rm, rv = torch.zeros(ch, device=DEV), torch.ones(ch, device=DEV)

    def bn(t, res=None):
        return ops.batch_norm_act(t, gam, bet, rm.clone(), rv.clone(), True, residual=res, relu=True)

    def build(fork, x, w0, w1, w2, gam, bet):
        h = ops.conv2d(x, w0, None, 1, 1)
        a, b = fork(h, 2)                        # bn2's residual share is written first, conv1's dgrad adds into it
        return bn(ops.conv2d(bn(ops.conv2d(a, w1, None, 1, 1)), w2, None, 1, 1), res=b)

    leaves = [x, w0, w1, w2, gam, bet]
    got = _run(build, leaves, True)
    want = _run(build, leaves, False)
    for a, b in zip(got, want):
        _close(a, b)


@pytest.mark.parametrize('shape', [(2, 16, 16, 20, 20, 3), (2, 3, 16, 20, 20, 7), (2, 64, 32, 12, 12, 3)])
def test_backward_data_add_with_the_output_as_addend(shape):
    """C ABI: `addend == grad_x` (and `addend2 == grad_x`) means grad_x += dgrad, whichever kernel family takes the layer."""
    import hip_runtime as hr
    from hip_runtime import ops
    B, C, Co, H, W, k = shape
    L = hr.lib()
    g = torch.Generator().manual_seed(7)
    gy = torch.randn(B, Co, H, W, generator=g).to(DEV)
    w = (torch.randn(Co, C, k, k, generator=g) / k).to(DEV)
    base = torch.randn(B, C, H, W, generator=g).to(DEV)
    geom = (B, C, H, W, Co, k, k, 1, 1, k // 2, k // 2)
    wp, wn = ops._ws(L.cnuda_conv2d_workspace_bytes(*geom), gy)
    plain = torch.empty_like(base)
    hr.check(L.cnuda_conv2d_backward_data(hr.ptr(gy), hr.ptr(w), hr.ptr(plain), *geom, wp, wn, hr.stream()), 'dgrad')
    for which in (0, 1):
        acc = base.clone()
        add = (hr.ptr(acc), hr.ptr(None)) if which == 0 else (hr.ptr(None), hr.ptr(acc))
        hr.check(L.cnuda_conv2d_backward_data_add(hr.ptr(gy), hr.ptr(w), add[0], add[1], hr.ptr(acc), *geom, wp, wn,
                                                  hr.stream()), 'dgrad_add')
        _close(acc, base + plain, 2e-6)
    other = torch.randn(B, C, H, W, generator=g).to(DEV)
    acc = base.clone()
    hr.check(L.cnuda_conv2d_backward_data_add(hr.ptr(gy), hr.ptr(w), hr.ptr(acc), hr.ptr(other), hr.ptr(acc), *geom, wp, wn,
                                              hr.stream()), 'dgrad_add2')
    _close(acc, base + other + plain, 2e-6)


def test_three_levels_of_forks_with_several_convolutions_on_the_innermost():
    """X forked from an alias of Y forked from an alias of Z: X merges into Y, Y into Z, then ANOTHER convolution of X arrives
    -- its slot must add to the live total (Z's buffer), and what X and Y held is counted exactly once."""
    from hip_runtime import ops
    x, w1, w2, w3, w4 = _leaves((2, 32, 12, 12), (32, 32, 3, 3), (32, 32, 3, 3), (32, 32, 3, 3), (32, 32, 1, 1))

    def build(fork, x, w1, w2, w3, w4):
        h = ops.conv2d(x, w1, None, 1, 1)
        z1, z2 = fork(h, 2)                      # Z: outermost
        y1, y2 = fork(z1, 2)                     # Y
        x1, x2, x3 = fork(y1, 3)                 # X: innermost, three convolutions
        # backward order = reverse of creation: z2's consumer first (owned buffer in Z), then x3, y2, x2, x1
        o = ops.conv2d(x1, w2, None, 1, 1)
        o = o + ops.conv2d(x2, w3, None, 1, 1)
        o = o + ops.conv2d(y2, w4, None, 1, 0)
        o = o + ops.conv2d(x3, w2, None, 1, 1)
        return o + ops.conv2d(z2, w3, None, 1, 1)

    leaves = [x, w1, w2, w3, w4]
    got = _run(build, leaves, True)
    want = _run(build, leaves, False)
    for a, b in zip(got, want):
        _close(a, b, 5e-6)
    got2 = _run(build, leaves, True)             # slots are per forward: a second build starts unmerged
    for a, b in zip(got2, got):
        assert torch.equal(a, b)
