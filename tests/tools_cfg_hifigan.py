"""Architecture of the small causal HiFi-GAN generator behind tests/golden/hifigan_small.npz (same values as CFG in
tools/make_golden_hifigan.py; the golden holds the weights)."""
CFG = dict(in_channels=16, out_channels=1, channels=32, kernel_size=7, upsample_scales=(4, 2), upsample_kernel_sizes=(8, 4),
           resblock_kernel_sizes=(3, 7, 11), resblock_dilations=[(1, 3, 5), (1, 3, 5), (1, 3, 5)])
