"""Command-line flags and the YAML-serialisable Config base (reference VarAutoEncoder/config.py).

The flag set is the reference's, name for name and default for default (config.py:19-70), because
scripts/train-vae.sh passes them; unknown flags are ignored as there (parse_known_args, :73-75). The flags
are declared as a table. Two additions, both off by default: --pianoroll (attach the piano-roll ends) and
--dtype."""
import argparse
import copy
import inspect

import yaml


def str2bool(v):
    return v.lower() in ("true", "1")


# (group, flag(s), kwargs) — reference config.py:19-70
_FLAGS = [
    ("Network", ("--e-n-layers",), dict(type=int, default=1)),
    ("Network", ("--e-rnn-hidden-dim",), dict(type=int, default=128)),
    ("Network", ("--e-emb-hidden-dim",), dict(type=int, default=64)),
    ("Network", ("--e-dropout",), dict(type=float, default=0.0)),
    ("Network", ("--e-num-heads",), dict(type=int, default=8)),
    ("Network", ("--latent-dim",), dict(type=int, default=64)),
    ("Network", ("--d-n-layers",), dict(type=int, default=1)),
    ("Network", ("--d-rnn-hidden-dim",), dict(type=int, default=128)),
    ("Network", ("--d-dropout",), dict(type=float, default=0.0)),
    ("Data", ("--batch-size",), dict(type=int, default=1)),
    ("Data", ("--max-seq-len",), dict(type=int, default=64)),
    ("Data", ("--slices-per-quarter-note",), dict(type=float, default=4)),
    ("Data", ("--data",), dict(type=str, default="data")),
    ("Data", ("--validation-data",), dict(type=str, default=None)),
    ("Data", ("--minimum-pattern-length",), dict(type=int, default=16)),
    ("Data", ("--pattern-identifier",), dict(type=str, choices=["recurring", ""], default="")),
    ("Training", ("--epochs",), dict(type=int, default=5000)),
    ("Training", ("--learning-rate",), dict(type=float, default=3e-4)),
    ("Training", ("--optimizer",), dict(type=str, default="adam")),
    ("Training", ("--optimizer-params",), dict(type=str, default="")),
    ("Training", ("--validation-split",), dict(type=float, default=0.1)),
    ("Training", ("--kl-loss",), dict(type=float, default=1.0)),
    ("Training", ("--label-smoothing",), dict(type=float, default=0.0)),
    ("Training", ("--negative-label-downscaling",), dict(action="store_true")),
    ("Training", ("--beam-size",), dict(type=int, default=5)),
    ("Training", ("--sampling-type",), dict(choices=["beam-search", "sampling"], default="sampling")),
    ("Misc", ("--load-checkpoint",), dict(type=int, default=1)),
    ("Misc", ("--checkpoint-frequency",), dict(type=int, default=5000)),
    ("Misc", ("--sampling-frequency",), dict(type=int, default=1000)),
    ("Misc", ("--num-checkpoints-not-improved",), dict(type=int, default=10)),
    ("Misc", ("--out-samples", "-o"), dict(type=str, default=None)),
    ("Misc", ("--model-output", "-m"), dict(type=str, default="models")),
    ("Misc", ("--checkpoint", "-c"), dict(type=int, default=-1)),
    ("Misc", ("--gpu",), dict(action="store_true")),
    ("Misc", ("--toy",), dict(action="store_true")),
    ("Misc", ("--visualize-samples",), dict(action="store_true")),
    ("Misc", ("--verbose",), dict(action="store_true")),
    # additions of this implementation (not in the reference)
    ("MI355X", ("--d-num-heads",), dict(type=int, default=None)),  # the reference leaves decoder heads unspecified
    ("MI355X", ("--pianoroll",), dict(action="store_true")),
    ("MI355X", ("--dtype",), dict(choices=["bf16", "fp16"], default="bf16")),
    ("MI355X", ("--max-steps",), dict(type=int, default=0)),
]


def build_parser():
    parser = argparse.ArgumentParser()
    groups = {}
    for group, flags, kw in _FLAGS:
        g = groups.get(group) or groups.setdefault(group, parser.add_argument_group(group))
        g.add_argument(*flags, **kw)
    return parser


parser = build_parser()


def get_config(argv=None):
    config, _unparsed = parser.parse_known_args(argv)
    return config


class _Tagged(yaml.YAMLObjectMetaclass):
    """every Config subclass gets the YAML tag !<ClassName> (config.py:81-87)"""

    def __init__(cls, name, bases, kwds):
        cls.yaml_tag = "!" + name
        kwds = dict(kwds, yaml_tag="!" + name)
        super().__init__(name, bases, kwds)


class Config(yaml.YAMLObject, metaclass=_Tagged):
    """Freezable, YAML (de-)serialisable configuration object (config.py:90-222). Loading uses an explicit
    loader (the reference's bare yaml.load fails on PyYAML >= 6)."""
    yaml_loader = yaml.UnsafeLoader

    def __init__(self):
        object.__setattr__(self, "_frozen", False)

    def __setattr__(self, key, value):
        if getattr(self, "_frozen", False):
            raise AttributeError("Cannot set '%s' in frozen config" % key)
        if value is self:
            raise AttributeError("Cannot set self as attribute")
        object.__setattr__(self, key, value)

    def __setstate__(self, state):
        self.__dict__.update(state)
        for name, param in inspect.signature(self.__init__).parameters.items():  # new args keep old files loadable
            if param.default is not param.empty and not hasattr(self, name):
                object.__setattr__(self, name, param.default)

    def _children(self):
        return [v for k, v in self.__dict__.items() if isinstance(v, Config) and k != "self"]

    def freeze(self):
        if getattr(self, "_frozen", False):
            return
        object.__setattr__(self, "_frozen", True)
        for c in self._children():
            c.freeze()

    def _strip_frozen(self):
        self.__dict__.pop("_frozen", None)
        for c in self._children():
            c._strip_frozen()

    def _add_frozen(self):
        object.__setattr__(self, "_frozen", False)
        for c in self._children():
            c._add_frozen()

    def __repr__(self):
        return "Config[%s]" % ", ".join("%s=%s" % (k, v) for k, v in sorted(self.__dict__.items()))

    def __eq__(self, other):
        if type(other) is not type(self):
            return False
        mine = {k: v for k, v in self.__dict__.items() if k not in ("self", "_frozen")}
        theirs = {k: v for k, v in other.__dict__.items() if k not in ("self", "_frozen")}
        return mine == theirs

    __hash__ = None

    def save(self, fname):
        obj = copy.deepcopy(self)
        obj._strip_frozen()
        with open(fname, "w") as out:
            yaml.dump(obj, out, default_flow_style=False)

    @staticmethod
    def load(fname):
        with open(fname) as inp:
            obj = yaml.load(inp, Loader=yaml.UnsafeLoader)
        obj._add_frozen()
        return obj

    def copy(self, **kwargs):
        c = copy.deepcopy(self)
        for k, v in kwargs.items():
            object.__setattr__(c, k, v)
        return c

    def output_to_stream(self, stream):
        for k, v in sorted(self.__dict__.items()):
            if k == "_frozen":
                continue
            if isinstance(v, Config):
                stream.write("%s:\n" % k)
                v.output_to_stream(stream)
            else:
                stream.write("  %s: %s\n" % (k, v))
