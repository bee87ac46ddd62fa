"""`with Quantize(model, optim, cfg.quant) as q: ...; q.convert()` (reference: pipeline/quant/context.py)."""
from .kmeans import KmeansQuant


class Quantize:
    def __init__(self, model, optim, quant_conf):
        self.model, self.optim, self.quant_conf = model, optim, quant_conf
        self.compress = None

    def __enter__(self):
        name = self.quant_conf["name"] if isinstance(self.quant_conf, dict) else self.quant_conf.name
        if name != "KMeans":
            raise NotImplementedError(f"quant '{name}': only KMeans is provided (QAT needs torch.quantization modules)")
        get = self.quant_conf.get
        skip_ll = get("skip_ll", ["layers.0.linear", "layers.7.linear"])
        self.compress = KmeansQuant(self.model, self.optim, bits=get("bits"), skip_ll=skip_ll)
        return self

    def __exit__(self, exc_type, exc, tb):
        return False

    def convert(self):
        self.compress.update_weights()
        return self.model
