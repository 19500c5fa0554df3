"""Image metrics used to report the PSNR parity criterion (definition of /root/reference/hugs/utils/image.py:27-29)."""
import torch


def psnr(img1, img2):
    """Per-image PSNR for images in [0,1]: 20 log10(1 / sqrt(mse)); inputs [B,C,H,W] -> [B,1]."""
    mse = ((img1 - img2) ** 2).reshape(img1.shape[0], -1).mean(1, keepdim=True)
    return 20.0 * torch.log10(1.0 / torch.sqrt(mse))
