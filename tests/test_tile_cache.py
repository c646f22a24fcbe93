"""Autotune tile cache: a file written against another library version / tile table must be ignored."""
import json

from animate_anything_amd import ops


def test_tile_cache_round_trip_and_version_check(emu, tmp_path):
    saved = dict(ops._tile_cache)
    try:
        ops._tile_cache.clear()
        ops._tile_cache[(0, 34, 64, 64, 64, 64, 64, 64, 1, 3, 3, 320, 0, 320, 0, False)] = (14, 0)
        path = str(tmp_path / "tiles.json")
        ops.save_tile_cache(path)
        ops._tile_cache.clear()
        assert ops.load_tile_cache(path) is True
        assert ops._tile_cache[(0, 34, 64, 64, 64, 64, 64, 64, 1, 3, 3, 320, 0, 320, 0, False)] == (14, 0)
        blob = json.load(open(path))
        blob["id"]["tiles"][3][0] += 64                     # a different tile table: indices mean something else
        json.dump(blob, open(path, "w"))
        ops._tile_cache.clear()
        assert ops.load_tile_cache(path) is False and not ops._tile_cache
        json.dump([[[1, 2], [3, 0]]], open(path, "w"))      # the round-1 format (no id) is rejected too
        assert ops.load_tile_cache(path) is False
    finally:
        ops._tile_cache.clear()
        ops._tile_cache.update(saved)
