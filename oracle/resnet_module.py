"""torchvision-shaped ResNet34 as a plain PyTorch-CPU nn.Module.  TEST INFRASTRUCTURE ONLY.

torchvision is not installed here, and the reference model takes its encoder as a constructor
argument (/root/reference/python/niantic/testing/test.py:151,161).  ``tests/golden/make_golden.py``
hands this module to the *reference's own* ``PoseNetX_R2`` class so that the reference code can run
on CPU; its state-dict keys are the torchvision 0.9.1 ones (conv1, bn1, layer{1..4}.{i}.conv{1,2},
bn{1,2}, downsample.{0,1}, fc).  The functional oracle in ``posenet_ref.py`` is checked against it.
"""
import torch
import torch.nn as nn


class _Block(nn.Module):
    def __init__(self, cin, cout, stride):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(cout)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(cout)
        self.downsample = None
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), nn.BatchNorm2d(cout))

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        y = self.bn2(self.conv2(self.relu(self.bn1(self.conv1(x)))))
        return self.relu(y + idt)


class ResNetCPU(nn.Module):
    def __init__(self, blocks=(3, 4, 6, 3), planes=(64, 128, 256, 512), num_classes=1000):
        super().__init__()
        self.conv1 = nn.Conv2d(3, planes[0], 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(planes[0])
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        cin = planes[0]
        for li, (c, nb) in enumerate(zip(planes, blocks), start=1):
            layers = []
            for bi in range(nb):
                layers.append(_Block(cin, c, 2 if (li > 1 and bi == 0) else 1))
                cin = c
            setattr(self, f"layer{li}", nn.Sequential(*layers))
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(cin, num_classes)

    def forward(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        return self.fc(torch.flatten(self.avgpool(x), 1))
