"""``instantiate_from_config`` with the reference's semantics (videotuna/utils/common_utils.py:90-109): a node is
``{target: dotted.path, params: {...}}``; targets under ``diffusers`` / ``transformers`` (or ``use_from_pretrained``)
are built with ``Class.from_pretrained(**params)``.  Third-party targets that the reference's YAMLs name are mapped
onto this engine's classes so ``configs/004_cogvideox/*.yaml`` load unchanged.
"""
from __future__ import annotations

import importlib
import os
from typing import Any

TARGET_REMAP = {
    "diffusers.CogVideoXTransformer3DModel": "vt355.dit.CogVideoXTransformer3DModel",
    "diffusers.CogVideoXDPMScheduler": "vt355.scheduler.CogVideoXDPMScheduler",
    "diffusers.AutoencoderKLCogVideoX": "vt355.vae.CogVideoXVaeEncoder",
    "peft.LoraConfig": "vt355.lora.LoraConfig",
    "videotuna.models.cogvideo_hf.cogvideo_pl.CogVideoXWorkFlow": "vt355.workflow.CogVideoXWorkFlow",
    "videotuna.models.cogvideo_hf.cogvideo_i2v.CogVideoXI2V": "vt355.workflow.CogVideoXI2V",
    "videotuna.models.lvdm.modules.encoders.condition.FrozenT5Embedder": "vt355.t5.FrozenT5Embedder",
    "transformers.T5EncoderModel": "vt355.t5.T5EncoderModel",
    # VideoCrafter2 (configs/001_videocrafter2/*.yaml)
    "videotuna.models.lvdm.modules.networks.openaimodel3d.UNetModel": "vt355.unet.UNetModel",
    "videotuna.flow.videocrafter.VideocrafterFlow": "vt355.lvdm.VideocrafterFlow",
    "videotuna.models.lvdm.ddpm3d.LVDMFlow": "vt355.lvdm.LVDMFlow",
    "videotuna.schedulers.ddpm.LDDPM": "vt355.lvdm.LDDPM",
    "videotuna.schedulers.diffusion_schedulers.LDMScheduler": "vt355.lvdm.LDMScheduler",
    # OpenSora v1.0 (configs/003_opensora/opensorav10_256x256.yaml)
    "videotuna.models.opensora.models.stdit.stdit.STDiT_XL_2": "vt355.stdit.STDiT_XL_2",
    "videotuna.models.opensora.models.stdit.stdit.STDiT": "vt355.stdit.STDiT",
    "videotuna.models.opensora.models.iddpm3d.LatentDiffusion": "vt355.stdit.OpenSoraFlow",
    "videotuna.models.opensora.models.iddpm3d.OpenSoraScheduler": "vt355.stdit.OpenSoraScheduler",
    # HunyuanVideo (in-tree denoiser class; the shipped LoRA recipe's diffusers class has no key map here)
    "videotuna.models.hunyuan.hyvideo_t2v.modules.models.HYVideoDiffusionTransformer": "vt355.hunyuan.HYVideoDiffusionTransformer",
    "videotuna.models.hunyuan.hyvideo_t2v.hunyuanvideo.HunyuanVideoWorkFlow": "vt355.hunyuan.HunyuanVideoFlow",
}
_NON_CTOR_KEYS = ("load_dtype",)       # consumed by the workflow, not by the class (cogvideo_pl.py:125-132)


def get_obj_from_str(string: str):
    string = TARGET_REMAP.get(string, string)
    module, cls = string.rsplit(".", 1)
    return getattr(importlib.import_module(module), cls)


def instantiate_from_config(config: Any, resolve=False):
    if config is None:
        return None
    if "target" not in config:
        if config in ("__is_first_stage__", "__is_unconditional__"):
            return None
        raise KeyError("Expected key `target` to instantiate.")
    target = config["target"]
    params = dict(config.get("params", dict()) or {})
    use_fp = bool(config.get("use_from_pretrained", False)) or "diffusers" in target or target.startswith("transformers")
    cls = get_obj_from_str(target)
    for k in _NON_CTOR_KEYS:
        params.pop(k, None)
    if use_fp and hasattr(cls, "from_pretrained") and "pretrained_model_name_or_path" in params:
        path = params["pretrained_model_name_or_path"]
        root = os.path.join(path, params.get("subfolder") or "")
        if os.path.isdir(root):
            return cls.from_pretrained(**params)
        raise FileNotFoundError(f"{target}: local checkpoint directory {root!r} does not exist (no network access); "
                                f"give explicit constructor params instead of pretrained_model_name_or_path")
    return cls(**params)


def load_yaml(path: str, env: dict = None) -> dict:
    """A reference config file as plain dicts.  ``${NAME}`` placeholders (the shipped YAMLs use them for user paths, e.g.
    ``csv_path: ${YOUR_DATA_CSV_PATH}``, resolved by OmegaConf in the reference) are filled from ``env`` / os.environ and
    left as they are when unknown -- they only occur under ``data:``, which this engine does not build."""
    import re
    import yaml
    with open(path) as f:
        cfg = yaml.safe_load(f)
    table = dict(os.environ)
    table.update(env or {})

    def walk(o):
        if isinstance(o, dict):
            return {k: walk(v) for k, v in o.items()}
        if isinstance(o, list):
            return [walk(v) for v in o]
        if isinstance(o, str) and "${" in o:
            return re.sub(r"\$\{([^}]+)\}", lambda m: str(table.get(m.group(1), m.group(0))), o)
        return o
    return walk(cfg)
