"""aldi_torch_randperm_prefix reproduces torch.randperm(n)[:k] AND the generator state afterwards, bit for bit (CPU only)."""
import pytest
import torch


@pytest.mark.parametrize("seed", [0, 7, 2**31 + 5])
def test_randperm_prefix_is_torch_randperm(seed):
    from aldi_amd.engine import randperm_prefix
    import aldi_amd.engine as E
    cases = [(0, 4), (1, 1), (2, 2), (3, 1), (623, 9), (624, 624), (625, 3), (1000, 128), (159882, 256), (268569, 256), (268569, 0), (1132, 512)]
    for n, k in cases:
        torch.manual_seed(seed)
        torch.randperm(91)                      # shift the position inside the 624-word block
        a = torch.randperm(n)[:k]
        tail_a = (torch.randperm(33), torch.rand(3))
        torch.manual_seed(seed)
        torch.randperm(91)
        b = randperm_prefix(n, k)
        tail_b = (torch.randperm(33), torch.rand(3))
        assert E._FAST_RANDPERM is True
        assert torch.equal(a, b), (n, k)
        assert torch.equal(tail_a[0], tail_b[0]) and torch.equal(tail_a[1], tail_b[1]), (n, k)
