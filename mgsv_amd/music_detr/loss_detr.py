"""Loss weights of the set criterion (reference music_detr/loss_detr.py:36-45)."""


def weight_dict(cfg):
    wd = {"loss_span": 4, "loss_giou": 1, "loss_label": 0.8}
    if cfg.contrastive_align_loss:
        wd["loss_contrastive_align"] = 0.2
    if cfg.aux_loss:
        base = dict(wd)
        for i in range(cfg.detr_dec_layers - 1):
            wd.update({f"{k}_{i}": v for k, v in base.items()})
    return wd
