"""Stock-torch members of the reference's model factory that are NOT on the MI355X hot path (SimpleCNN, and -- until
its attention kernels exist -- cnn_transformer).

``SimpleCNN`` is BASELINE.json configs[0] ("SimpleCNN ... CPU PyTorch reference -- plumbing, no GPU"): the starter
ResNet-style CNN of reference src/models.py:44-123.  It stays a plain ``nn.Module`` built from stock layers (it runs
wherever torch runs, CPU included), restated here only so that ``get_model`` serves every ``model.type`` of the
reference; attribute names and registration order follow the reference so that ``state_dict`` keys, shapes and the
default initialisation under ``torch.manual_seed`` are identical (tests/test_host_cpu.py pins them against fixtures
generated from the reference).
"""
import torch
import torch.nn as nn


class ResidualBlock(nn.Module):
    """conv-BN-ReLU-conv-BN + (1x1 conv-BN | identity) skip, ReLU  (reference src/models.py:44-76)."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1):
        super().__init__()
        pad = kernel_size // 2
        self.conv1 = nn.Conv2d(in_channels, out_channels, kernel_size, stride=stride, padding=pad)
        self.bn1 = nn.BatchNorm2d(out_channels)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(out_channels, out_channels, kernel_size, padding=pad)
        self.bn2 = nn.BatchNorm2d(out_channels)
        self.skip = nn.Sequential()
        if stride != 1 or in_channels != out_channels:
            self.skip = nn.Sequential(nn.Conv2d(in_channels, out_channels, kernel_size=1, stride=stride),
                                      nn.BatchNorm2d(out_channels))

    def forward(self, x):
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        out = out + self.skip(x)
        return self.relu(out)


class SimpleCNN(nn.Module):
    """Reference src/models.py:79-123: stem -> ``depth`` residual blocks (channels double until the last) -> Dropout2d
    -> conv-BN-ReLU -> 1x1 conv.  x [B, n_input_channels, H, W] -> [B, n_output_channels, H, W]."""

    def __init__(self, n_input_channels, n_output_channels, kernel_size=3, init_dim=64, depth=4, dropout_rate=0.2):
        super().__init__()
        pad = kernel_size // 2
        self.initial = nn.Sequential(nn.Conv2d(n_input_channels, init_dim, kernel_size=kernel_size, padding=pad),
                                     nn.BatchNorm2d(init_dim), nn.ReLU(inplace=True))
        self.res_blocks = nn.ModuleList()
        dim = init_dim
        for i in range(depth):
            last = i == depth - 1
            self.res_blocks.append(ResidualBlock(dim, dim if last else dim * 2))
            if not last:
                dim *= 2
        self.dropout = nn.Dropout2d(dropout_rate)
        self.final = nn.Sequential(nn.Conv2d(dim, dim // 2, kernel_size=kernel_size, padding=pad),
                                   nn.BatchNorm2d(dim // 2), nn.ReLU(inplace=True),
                                   nn.Conv2d(dim // 2, n_output_channels, kernel_size=1))

    def forward(self, x):
        x = self.initial(x)
        for blk in self.res_blocks:
            x = blk(x)
        return self.final(self.dropout(x))


class CNNTransformer(nn.Module):
    """Reference src/cnn_transformer.py:4-54 (BASELINE.json configs[3]): two stride-2 convs (48x72 -> 12x18 = 216
    tokens), learned positional embedding, ``depth`` post-norm transformer encoder layers (ReLU MLP, dropout), two
    2x2 transposed convs and a 1x1 head.  STOCK TORCH for now: this model type is served so that the factory is
    complete and checkpoints interoperate, but it is not (yet) on the hand-written HIP path -- see DESIGN.md section 7.
    x [B, in_channels, 48, 72] -> [B, out_channels, 48, 72]."""

    def __init__(self, in_channels=5, out_channels=2, embed_dim=128, depth=4, n_heads=4, mlp_dim=256, dropout=0.1):
        super().__init__()
        half, quarter = embed_dim // 2, embed_dim // 4
        self.encoder = nn.Sequential(nn.Conv2d(in_channels, half, kernel_size=3, stride=2, padding=1), nn.ReLU(),
                                     nn.Conv2d(half, embed_dim, kernel_size=3, stride=2, padding=1), nn.ReLU())
        self.height, self.width = 12, 18
        self.num_tokens = self.height * self.width
        self.embed_dim = embed_dim
        self.pos_embedding = nn.Parameter(torch.randn(1, self.num_tokens, embed_dim))
        layer = nn.TransformerEncoderLayer(d_model=embed_dim, nhead=n_heads, dim_feedforward=mlp_dim, dropout=dropout,
                                           batch_first=True)
        self.transformer = nn.TransformerEncoder(layer, num_layers=depth)
        self.decoder = nn.Sequential(nn.ConvTranspose2d(embed_dim, half, kernel_size=2, stride=2), nn.ReLU(),
                                     nn.ConvTranspose2d(half, quarter, kernel_size=2, stride=2), nn.ReLU(),
                                     nn.Conv2d(quarter, out_channels, kernel_size=1))

    def forward(self, x):
        b = x.size(0)
        tok = self.encoder(x).flatten(2).transpose(1, 2) + self.pos_embedding        # [B, 216, E]
        tok = self.transformer(tok)
        return self.decoder(tok.transpose(1, 2).reshape(b, self.embed_dim, self.height, self.width))
