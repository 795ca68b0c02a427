"""hnet_group (include/hnet.h, round 6): independent steps issued round-robin on several contexts, each on its own stream - the results of every step are those of
a single context, bit for bit, whatever the number of members; the members' streams are distinct; join / synchronize order the caller's work behind the steps."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n_ctx", [2, 3, 4])
@pytest.mark.parametrize("variant,n_mc,batch", [("prior3", 16, 32), ("full", 32, 5)])
def test_group_steps_equal_single_context_steps(blob, n_ctx, variant, n_mc, batch):
    import torch
    from cuahn_vio_amd import synth
    from cuahn_vio_amd.homography_net import PIX_U8, HnetEngine, HnetGroup
    dev = torch.device("cuda:0")
    steps = 9
    ph, ch, prh, _ = synth.make_batch(900, batch)
    prev, curr, prior = torch.from_numpy(ph).to(dev), torch.from_numpy(ch).to(dev), torch.from_numpy(prh).to(dev)
    dp = prior.data_ptr() if variant != "full" else None
    kw = dict(variant=variant, mc_samples=n_mc, dropout_p=0.05, mc_seed=9, max_batch=batch)
    ref = torch.zeros(steps, batch, 72, device=dev)
    e = HnetEngine(blob, **kw)
    for i in range(steps):
        e.infer_batch_packed_device(prev.data_ptr(), curr.data_ptr(), PIX_U8, dp, batch, 1000 * i, ref[i].data_ptr())
    e.synchronize()
    e.close()
    g = HnetGroup(blob, n_ctx, **kw)
    assert g.n == n_ctx and len({g.stream(i) for i in range(n_ctx)}) == n_ctx and all(g.stream(i) for i in range(n_ctx))
    out = torch.zeros(steps, batch, 72, device=dev)
    used = [g.infer_batch_packed_device(prev.data_ptr(), curr.data_ptr(), PIX_U8, dp, batch, 1000 * i, out[i].data_ptr()) for i in range(steps)]
    assert used == [i % n_ctx for i in range(steps)]
    # a consumer stream joined behind the members reads complete results without a host synchronisation in between
    s = torch.cuda.Stream(dev)
    g.join(s)
    with torch.cuda.stream(s):
        total = out.sum()
    s.synchronize()
    torch.cuda.synchronize()
    assert g.overflow_flag() == 0
    assert torch.equal(out, ref) and float(total) == float(ref.sum())
    assert g.members[0].config().mc_samples == n_mc
    g.close()


def test_group_refuses_bad_sizes(blob):
    from cuahn_vio_amd.homography_net import HnetError, HnetGroup
    for n in (0, 9):
        with pytest.raises(HnetError):
            HnetGroup(blob, n)
