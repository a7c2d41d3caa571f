"""Command-line surface of train.py / test.py (reference: options/base_options.py:13-265,
options/train_options.py, options/test_options.py).  Same flags, same three parsing passes
(base -> model -> dataset), same post-parse fix-ups (model synonyms, gpu id list, SORTED
person/cloth inputs — the sort fixes the channel order —, n_frames_now default)."""
import argparse
import sys

from . import data, registry


class BaseOptions:
    def __init__(self):
        self.initialized = False
        self.is_train = None

    def initialize(self, parser):
        parser.add_argument("--name", default="unnamed_experiment")
        parser.add_argument("--distributed_backend", default="ddp", help="how to do distributed multigpu training")
        parser.add_argument("--gpu_ids", default="0", help="comma separated of which GPUs to train on")
        parser.add_argument("-j", "--num_workers", "--workers", dest="workers", type=int, default=4)
        parser.add_argument("-b", "--batch_size", type=int, default=8)
        parser.add_argument("--activation", choices=("relu", "gelu", "swish", "sine"))
        parser.add_argument("-fp", "--precision", type=int, dest="precision", choices=(16, 32), default=16,
                            help="accepted for compatibility; the MI355X path computes in fp32 (DESIGN.md)")
        parser.add_argument("--dataset", choices=("viton", "viton_vvt_mpv", "vvt", "mpv", "synthetic"), default="vvt")
        parser.add_argument("--datamode", default="train")
        parser.add_argument("--model", help="'warp' (aka 'gmm'), 'unet_mask' (aka 'tom', 'unet')")
        parser.add_argument("--datacap", "--datacap_train", "--limit_train_batches", dest="limit_train_batches",
                            default="1.0")
        parser.add_argument("--datacap_val", "--limit_val_batches", dest="limit_val_batches", default="1.0")
        parser.add_argument("--experiments_dir", default="experiments")
        parser.add_argument("--checkpoint", type=str, default="", help="model checkpoint for initialization")
        parser.add_argument("--display_count", type=int, default=200)
        parser.add_argument("--loglevel", choices=("debug", "info", "warning", "error", "critical"), default="info")
        parser.add_argument("--fast_dev_run", action="store_true")
        self.initialized = True
        return parser

    def gather_options(self, argv=None):
        parser = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter)
        parser = self.initialize(parser)
        opt, _ = parser.parse_known_args(argv)
        BaseOptions.apply_model_synonyms(opt)
        parser = registry.get_option_setter(opt.model)(parser, self.is_train)
        opt, _ = parser.parse_known_args(argv)
        parser = data.get_option_setter(opt.dataset)(parser, self.is_train)
        self.parser = parser
        return parser.parse_args(argv)

    def parse(self, argv=None, interactive=True):
        opt = self.gather_options(argv)
        opt.is_train = self.is_train
        if interactive:
            BaseOptions.apply_ask_unnamed_experiment(opt, argv)
        BaseOptions.apply_model_synonyms(opt)
        BaseOptions.apply_gpu_ids(opt)
        BaseOptions.apply_val_check_ge_train_batch(opt)
        BaseOptions.apply_sort_inputs(opt)
        if getattr(opt, "n_frames_now", None) is None:
            opt.n_frames_now = getattr(opt, "n_frames_total", 1)
        if hasattr(opt, "encoder_input") and opt.encoder_input is None:  # SamsModel.apply_default_encoder_input
            opt.encoder_input = opt.person_inputs[0]
        self.opt = opt
        return opt

    @staticmethod
    def apply_ask_unnamed_experiment(opt, argv=None):
        args = sys.argv if argv is None else argv
        if "--name" not in args and sys.stdin is not None and sys.stdin.isatty():
            new_name = input(f"Experiment name (default: {opt.name}): ")
            if new_name:
                opt.name = new_name

    @staticmethod
    def apply_gpu_ids(opt):
        if isinstance(opt.gpu_ids, str):
            opt.gpu_ids = [int(s) for s in opt.gpu_ids.split(",") if int(s) >= 0]

    @staticmethod
    def apply_model_synonyms(opt):
        if opt.model is None:
            raise SystemExit("--model is required ('warp' or 'unet_mask')")
        opt.model = registry.canonical_model_name(opt.model)

    @staticmethod
    def apply_sort_inputs(opt):
        opt.person_inputs = sorted(opt.person_inputs)
        opt.cloth_inputs = sorted(opt.cloth_inputs)

    @staticmethod
    def apply_val_check_ge_train_batch(opt):
        if hasattr(opt, "val_check_interval"):
            if opt.fast_dev_run:
                opt.val_check_interval = 1
                return
            v, lim = str2num(str(opt.val_check_interval)), str2num(str(opt.limit_train_batches))
            if isinstance(v, int) and isinstance(lim, int) and v > lim:
                opt.val_check_interval = opt.limit_train_batches


def str2num(s):
    try:
        return int(s)
    except ValueError:
        return float(s)


class TrainOptions(BaseOptions):
    def initialize(self, parser):
        parser = BaseOptions.initialize(self, parser)
        parser.add_argument("--no_shuffle", action="store_true", help="don't shuffle input data")
        parser.add_argument("--save_count", type=int, default=10000)
        parser.add_argument("--val_check_interval", "--val_frequency", dest="val_check_interval", type=str,
                            default="0.125")
        parser.add_argument("--lr", type=float, default=1e-4, help="initial learning rate for adam")
        parser.add_argument("--keep_epochs", type=int, default=5)
        parser.add_argument("--decay_epochs", type=int, default=5)
        parser.add_argument("--accumulated_batches", type=int, default=1)
        self.is_train = True
        return parser


class TestOptions(BaseOptions):
    def initialize(self, parser):
        parser = BaseOptions.initialize(self, parser)
        parser.add_argument("--no_shuffle", action="store_true", default=True)
        parser.set_defaults(datamode="test")
        self.is_train = False
        parser.add_argument("--result_dir", type=str, default="test_results", help="save test result outputs")
        parser.add_argument("--tryon_list", help="CSV with CLOTH_PATH and PERSON_ID columns")
        parser.add_argument("--random_tryon", action="store_true")
        return parser
