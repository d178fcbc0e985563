"""Checkpoint files: the flow's `state_dict` (the reference's only weight-transfer format, SURVEY 8b - its keys
are the reference's) plus a small metadata dict, written with torch.save and read back with
`weights_only=True` (nothing in the file is executed on load)."""
import torch

FORMAT = 'normalizing-flows-amd/1'


def save_checkpoint(module, path, **meta):
    """Writes {'format', 'state_dict', 'meta'}; `meta` values must be plain Python numbers / strings / lists."""
    for k, v in meta.items():
        if not isinstance(v, (int, float, str, bool, list, tuple, type(None))):
            raise TypeError('metadata %r must be a plain Python value, got %s' % (k, type(v).__name__))
    sd = {k: v.detach().cpu() for k, v in module.state_dict().items()}
    torch.save({'format': FORMAT, 'state_dict': sd, 'meta': dict(meta)}, path)


def load_checkpoint(module, path, strict=True, map_location='cpu'):
    """Loads a file written by save_checkpoint (or a bare reference `state_dict` file) into `module`;
    returns the metadata dict."""
    blob = torch.load(path, map_location=map_location, weights_only=True)
    if isinstance(blob, dict) and blob.get('format') == FORMAT:
        sd, meta = blob['state_dict'], blob.get('meta', {})
    elif isinstance(blob, dict) and all(isinstance(v, torch.Tensor) for v in blob.values()):
        sd, meta = blob, {}                         # a plain state_dict, e.g. saved from the reference
    else:
        raise ValueError('%s is not a %s checkpoint or a state_dict' % (path, FORMAT))
    module.load_state_dict(sd, strict=strict)
    return meta
