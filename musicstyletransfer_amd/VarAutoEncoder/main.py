"""Entry point: python -m music_style_transfer.VarAutoEncoder.main <flags> (scripts/train-vae.sh:5), wiring
flags -> Loader/Dataset -> ModelConfig/Model -> Trainer.fit exactly as the reference's main.py:122-172 does.
Differences, all forced: the decoder gets a TransformerConfig (the reference passes a non-existent
lstm_config kwarg and raises TypeError), --pianoroll selects the piano-roll ends, and a context without a HIP
device is refused loudly instead of running slowly on the CPU."""
import os

from . import model, trainer
from .config import get_config
from .data import Loader, ToyData, load_dataset
from .sampler import get_sampler
from .transformer import TransformerConfig
from .utils import cpu, create_directory_if_not_present, gpu, log_config


def create_toy_model_config(data):
    """main.py:14-38"""
    def t():
        return TransformerConfig(model_size=32, dropout=0.0, num_layers=1, vocab_size=data.num_tokens(), num_heads=2)
    return model.ModelConfig(
        encoder_config=model.EncoderConfig(transformer_config=t(), latent_dim=16, num_classes=data.num_classes(),
                                           input_dim=data.num_tokens()),
        decoder_config=model.DecoderConfig(transformer_config=t(), latent_dim=16, num_classes=data.num_classes(),
                                           output_dim=data.num_tokens()))


def create_toy_train_config(max_steps=0):
    """main.py:41-55"""
    return trainer.TrainConfig(batch_size=1, sampling_frequency=500, checkpoint_frequency=1000, num_checkpoints_not_improved=-1,
                               kl_loss=1.0, optimizer=trainer.OptimizerConfig(learning_rate=1e-3, optimizer="adam",
                                                                              optimizer_params="clip_gradient:1.0"),
                               label_smoothing=0.0, negative_label_downscaling=True, verbose=False, max_steps=max_steps)


def main_toy(context, args):
    dataset = ToyData()
    config = create_toy_model_config(dataset)
    m = model.Model(config=config)
    model_folder = os.path.join(args.model_output if args.model_output != "models" else "/tmp/music-style-transfer/toy", "model")
    create_directory_if_not_present(model_folder)
    config.save(os.path.join(model_folder, "config"))
    t = trainer.Trainer(config=create_toy_train_config(args.max_steps), context=context, model=m, sampler=None)
    t.fit(dataset=dataset, validation_dataset=dataset, model_folder=model_folder, epochs=20000 if not args.max_steps else args.max_steps)
    return t


def create_train_config(args):
    """main.py:79-93"""
    return trainer.TrainConfig(batch_size=args.batch_size, sampling_frequency=args.sampling_frequency,
                               checkpoint_frequency=args.checkpoint_frequency,
                               num_checkpoints_not_improved=args.num_checkpoints_not_improved, kl_loss=args.kl_loss,
                               optimizer=trainer.OptimizerConfig(learning_rate=args.learning_rate, optimizer=args.optimizer,
                                                                 optimizer_params=args.optimizer_params),
                               label_smoothing=args.label_smoothing, negative_label_downscaling=args.negative_label_downscaling,
                               verbose=args.verbose, dtype=args.dtype, max_steps=args.max_steps)


def create_model_config(args, dataset):
    """main.py:96-118 with the decoder given the TransformerConfig that model.DecoderConfig requires"""
    d_heads = args.d_num_heads or args.e_num_heads
    return model.ModelConfig(
        encoder_config=model.EncoderConfig(
            transformer_config=TransformerConfig(model_size=args.e_rnn_hidden_dim, dropout=args.e_dropout, num_layers=args.e_n_layers,
                                                 vocab_size=dataset.num_tokens(), num_heads=args.e_num_heads),
            latent_dim=args.latent_dim, num_classes=dataset.num_classes(), input_dim=dataset.num_tokens()),
        decoder_config=model.DecoderConfig(
            transformer_config=TransformerConfig(model_size=args.d_rnn_hidden_dim, dropout=args.d_dropout, num_layers=args.d_n_layers,
                                                 vocab_size=dataset.num_tokens(), num_heads=d_heads),
            latent_dim=args.latent_dim, num_classes=dataset.num_classes(), output_dim=dataset.num_tokens()),
        kind="pianoroll" if args.pianoroll else "token")


def main(argv=None):
    args = get_config(argv)
    context = gpu() if args.gpu else cpu()
    if args.toy:
        return main_toy(context, args)
    loader = Loader(path=args.data, max_sequence_length=args.max_seq_len, slices_per_quarter_note=args.slices_per_quarter_note)
    val_loader = (Loader(path=args.validation_data, max_sequence_length=args.max_seq_len,
                         slices_per_quarter_note=args.slices_per_quarter_note) if args.validation_data is not None else None)
    kw = {}
    if args.pianoroll:
        from ..pianoroll import PianoRollDataset
        kw = dict(dataset_cls=lambda bs, L, mel, **k: PianoRollDataset(bs, L, mel, slices_per_quarter=args.slices_per_quarter_note))
    train_dataset, valid_dataset = load_dataset(loader, args.batch_size, args.validation_split, val_loader, **kw)
    create_directory_if_not_present(args.model_output)
    create_directory_if_not_present(args.out_samples)
    config = create_model_config(args, train_dataset)
    config.save(os.path.join(args.model_output, "config"))
    log_config(config)
    m = model.Model(config=config)
    sampler = get_sampler("sampling", args.model_output, context, None, args)
    t = trainer.Trainer(config=create_train_config(args), context=context, model=m, sampler=sampler)
    t.fit(dataset=train_dataset, validation_dataset=valid_dataset, model_folder=args.model_output, epochs=args.epochs)
    print("Training finished.")
    return t


if __name__ == "__main__":
    main()
