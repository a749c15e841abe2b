"""SimSiam head on the deepest encoder feature (reference MD2/contrastive.py:6-93): global average pool,
3-layer projector, 2-layer predictor, symmetric negative cosine similarity with stop-gradient targets.
Tiny GEMMs: stays on PyTorch; its 2.06 M parameters join the gradient all-reduce."""
import torch.nn as nn


class SimSiam(nn.Module):
    def __init__(self, dim=1000, pred_dim=512):
        super().__init__()
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.my_cos = nn.CosineSimilarity()
        prev_dim = 512
        self.projector = nn.Sequential(nn.Linear(prev_dim, prev_dim, bias=False), nn.BatchNorm1d(prev_dim),
                                       nn.ReLU(inplace=True),
                                       nn.Linear(prev_dim, prev_dim, bias=False), nn.BatchNorm1d(prev_dim),
                                       nn.ReLU(inplace=True),
                                       nn.Linear(prev_dim, dim, bias=False), nn.BatchNorm1d(dim, affine=False))
        self.predictor = nn.Sequential(nn.Linear(dim, pred_dim, bias=False), nn.BatchNorm1d(pred_dim),
                                       nn.ReLU(inplace=True), nn.Linear(pred_dim, dim))

    def forward(self, feature1, feature2):
        """feature1/feature2: encoder feature lists of the adversarial and the benign view."""
        z1 = self.avgpool(feature1[-1]).flatten(1)
        z2 = self.avgpool(feature2[-1]).flatten(1)
        z1, z2 = self.projector(z1), self.projector(z2)
        p1, p2 = self.predictor(z1), self.predictor(z2)
        z1, z2 = z1.detach(), z2.detach()
        return -(self.my_cos(p1, z2).mean() + self.my_cos(p2, z1).mean()) * 0.5
