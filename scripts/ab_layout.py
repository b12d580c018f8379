"""A/B across the cell-word layout change (ABI 5 -> 6): libraries of ABI <= 5 read the 8x8-tile order"""
import numpy as np


def tiled_world(world):
    """the same world with the cell words in the 8x8-tile order of ABI <= 5 libraries"""
    from torchdriveenv_amd.world import World
    arrays = {k: v.copy() for k, v in world.arrays.items()}
    words, base = [], 0
    for m in arrays["maps"]:
        nx, ny, pitch = int(m["nx"]), int(m["ny"]), 1 << int(m["row_shift"])
        a = world.arrays["cell_word"][m["cell_base"]:m["cell_base"] + pitch * ny].reshape(ny, pitch)[:, :nx]
        words.append(np.ascontiguousarray(a.reshape(ny // 8, 8, nx // 8, 8).transpose(0, 2, 1, 3)).reshape(-1))
        m["cell_base"] = base
        base += nx * ny
    arrays["cell_word"] = np.concatenate(words)
    return World(arrays, world.ints, threshold=world.threshold)
