"""GUMBEL_SOFTMAX training branch of Quantize (reference modules/quantize.py:125-130 + distributions/gumbel.py:8-18).

No shipped config selects it (both gin files bind ROTATION_TRICK) although it is the HRqVae constructor default; the
eval branch (used by the tokenizer) never reaches it.  The fused RQ kernel covers STE / ROTATION / eval; the Gumbel
training branch is not built yet and fails loudly rather than falling back to anything else."""


def _fail():
    raise NotImplementedError("QuantizeForwardMode.GUMBEL_SOFTMAX in TRAINING mode is not built on the HIP path yet "
                              "(STE, ROTATION_TRICK and every eval-mode call are); see DESIGN.md, 'out of scope this round'")


def gumbel_level(layer, x, temperature):
    _fail()


def gumbel_all_levels(model, y, normalize_input):
    _fail()
