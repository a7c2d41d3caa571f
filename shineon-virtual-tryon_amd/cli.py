"""`python train.py ...` / `python test.py ...`: the reference's two entry points (train.py:32-141, test.py:9-10) over this
package's options, registry and Trainer.

    parse (TrainOptions | TestOptions)  ->  find_model_using_name(opt.model)
    --checkpoint given:  Model.load_from_checkpoint(path)   else:  Model(opt)
    model.override_hparams(opt)   (test-time / non-architectural flags win over the checkpoint's, base_model.py:76-89)
    Trainer(resume_from_checkpoint=opt.checkpoint or None, <hardware kwargs>, <train kwargs>)
    train: trainer.fit(model)   - Ctrl-C writes checkpoints/interrupted_by_Ctrl-C.ckpt, any exception writes
                                  checkpoints/interrupted_by_<ExceptionName>.ckpt (train.py:121-137; both inside Trainer.fit,
                                  at a step boundary with the streams drained) and the process exits NON-ZERO
    test:  trainer.test(model)  - test_step writes the PNGs the next stage reads

One process per GPU: under `python -m torch.distributed.run --nproc-per-node N train.py ...` every rank runs this main() and
Trainer joins the RCCL group from RANK / WORLD_SIZE (the reference lets Lightning spawn its DDP children from --gpu_ids).
"""
import logging
import os.path as osp
import sys
import traceback

import torch

from .options import TestOptions, TrainOptions, str2num
from .registry import find_model_using_name

logger = logging.getLogger("logger")
SEED = 420   # train.py:29 - DDP needs every rank to build the same initial weights


def hardware_kwargs(opt):
    return {"gpus": opt.gpu_ids, "distributed_backend": opt.distributed_backend, "precision": opt.precision}


def train_kwargs(opt):
    """Trainer arguments that only exist when training (train.py:89-118)."""
    if not opt.is_train:
        return {}
    return {
        "default_root_dir": osp.join(opt.experiments_dir, opt.name),
        "save_count": opt.save_count,                      # CheckpointEveryNSteps(opt.save_count)
        "accumulate_grad_batches": opt.accumulated_batches,
        "max_epochs": opt.keep_epochs + opt.decay_epochs,
        "val_check_interval": str2num(str(opt.val_check_interval)),
        "limit_train_batches": str2num(str(opt.limit_train_batches)),
        "limit_val_batches": str2num(str(opt.limit_val_batches)),
        "fast_dev_run": opt.fast_dev_run,
    }


def build_model(opt):
    """New model from the options, or the checkpoint's model with the command line's hparams laid over it."""
    model_class = find_model_using_name(opt.model)
    if opt.checkpoint:
        model = model_class.load_from_checkpoint(opt.checkpoint)
        logger.info("RESUMED %s from checkpoint: %s", model_class.__name__, opt.checkpoint)
    else:
        model = model_class(opt)
        logger.info("INITIALIZED new %s", model_class.__name__)
    model.override_hparams(opt)
    return model


def main(train=True, argv=None):
    from .trainer import Trainer

    torch.manual_seed(SEED)
    opt = (TrainOptions() if train else TestOptions()).parse(argv)
    logging.basicConfig(format="%(asctime)s %(levelname)s %(message)s")
    logger.setLevel(getattr(logging, opt.loglevel.upper()))
    model = build_model(opt)
    trainer = Trainer(resume_from_checkpoint=opt.checkpoint or None, **hardware_kwargs(opt), **train_kwargs(opt))
    if train:
        try:
            trainer.fit(model)
        except Exception as e:  # noqa: BLE001 - Trainer.fit has already written interrupted_by_<name>.ckpt
            logger.warning("Caught a %s!", type(e))
            logger.error(traceback.format_exc())
            return 1
    else:
        print("Testing........")
        print(opt)
        trainer.test(model)
    logger.info("Finished %s, named %s!", opt.model, opt.name)
    return 0


def run(train):
    sys.exit(main(train=train))
