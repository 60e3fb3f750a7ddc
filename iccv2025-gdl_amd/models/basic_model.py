"""Drop-in mirror of /root/reference/models/basic_model.py (`AVClassifier_DGL`, :10-86).

Same constructor (reads args.fusion_method / dataset / modality), same attribute and parameter
names in the same registration order (fusion head, audio_net, visual_net), same forward
signature and return order `(out, a_out, v_out)`.  The two encoders run concurrently on two
HIP streams.  All four DGL fusion heads of the reference are built (`concat`, `sum`, `gated` with
x_gate=True, `film`; fusion_modules.py:16-30,45-59,126-178,213-250) for the full-modality setting of the
DGL scripts; `modality != 'full'` raises NotImplementedError.  FiLM_DGL handles at most 64 samples per
call (its kernels put one sample per lane of a wavefront; the reference scripts train with batch 64).
"""
import torch
import torch.nn as nn

from .backbone import resnet18
from .fusion_modules import ConcatFusion, ConcatFusion_DGL, FiLM_DGL, GatedFusion_DGL, SumFusion_DGL  # noqa: F401

N_CLASSES = {'VGGSound': 309, 'KineticSound': 34, 'kinect400': 400, 'CREMAD': 6, 'AVE': 28}  # basic_model.py:15-26


class AVClassifier_DGL(nn.Module):
    def __init__(self, args):
        super(AVClassifier_DGL, self).__init__()
        fusion = args.fusion_method
        if args.dataset not in N_CLASSES:
            raise NotImplementedError('Incorrect dataset name {}'.format(args.dataset))
        n_classes = N_CLASSES[args.dataset]
        if fusion == 'sum':
            self.fusion_module = SumFusion_DGL(output_dim=n_classes)
        elif fusion == 'concat':
            self.fusion_module = ConcatFusion_DGL(output_dim=n_classes)
        elif fusion == 'gated':
            self.fusion_module = GatedFusion_DGL(output_dim=n_classes, x_gate=True)
        elif fusion == 'film':
            self.fusion_module = FiLM_DGL(output_dim=n_classes, x_film=True)
        else:
            raise NotImplementedError('Incorrect fusion method: {}!'.format(fusion))
        if args.modality != 'full':
            raise NotImplementedError("gdl: only modality='full' (the DGL scripts' setting) is implemented")
        self.audio_net = resnet18(modality='audio', args=args)
        self.visual_net = resnet18(modality='visual', args=args)
        self.modality = args.modality
        self.args = args
        self._side = None

    def forward(self, audio, visual):
        cur = torch.cuda.current_stream(audio.device)
        if self._side is None or self._side.device != audio.device:
            self._side = torch.cuda.Stream(device=audio.device)
        side = self._side
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            a = self.audio_net.forward_pooled(audio)  # [B,512]
        v = self.visual_net.forward_pooled(visual)  # [B,512]
        cur.wait_stream(side)
        a.record_stream(cur)
        a_out, v_out, out = self.fusion_module(a, v)
        return out, a_out, v_out


class AVClassifier(nn.Module):
    """BASELINE config 1: the non-DGL concat model of main.py.  The reference class of this name
    no longer exists in models/basic_model.py (main.py:19 cannot be imported); this
    restates its math from the parts that do exist: the two encoders, the pooling glue of
    basic_model.py:73-82 and `ConcatFusion` (fusion_modules.py:33-42).  Returns (a, v, out)."""

    def __init__(self, args):
        super(AVClassifier, self).__init__()
        if args.dataset not in N_CLASSES:
            raise NotImplementedError('Incorrect dataset name {}'.format(args.dataset))
        if args.fusion_method != 'concat':
            raise NotImplementedError('gdl: fusion method {!r} is not built yet (concat only)'.format(args.fusion_method))
        self.fusion_module = ConcatFusion(output_dim=N_CLASSES[args.dataset])
        self.audio_net = resnet18(modality='audio', args=args)
        self.visual_net = resnet18(modality='visual', args=args)
        self.modality = 'full'
        self.args = args

    def forward(self, audio, visual):
        a = self.audio_net.forward_pooled(audio)
        v = self.visual_net.forward_pooled(visual)
        a, v, out = self.fusion_module(a, v)
        return a, v, out
