"""The step's input contract (reference aldi/dataloader.py:45-80) plus a synthetic loader.

``unpack_data_weak_strong`` has the reference's semantics: returns
(labeled_weak, labeled_strong, unlabeled_weak, unlabeled_strong); weak = deepcopy with
"image" <- "img_weak"; unlabeled_weak is produced whenever ANY unlabeled content is requested."""
import copy

import torch

from . import synthetic

WEAK_IMG_KEY = "img_weak"


def _weak_view(batch):
    """independent copy of a batch whose "image" is the weakly augmented view (when the mapper attached one)"""
    out = []
    for rec in batch:
        rec = copy.deepcopy(rec)
        weak = rec.get(WEAK_IMG_KEY)
        if weak is not None:
            rec["image"] = weak
        out.append(rec)
    return out


def unpack_data_weak_strong(labeled, unlabeled, batch_contents=("labeled_weak", "labeled_strong", "unlabeled_strong")):
    """-> (labeled_weak, labeled_strong, unlabeled_weak, unlabeled_strong), None for what `batch_contents` does not ask for
    (reference aldi/dataloader.py:57-80; pinned by golden g7).  The strong entries are the incoming lists themselves; the weak
    ones are copies.  The unlabeled weak view also exists whenever the strong one is requested: the teacher labels it."""
    wanted = set(batch_contents)
    views = {"labeled": (labeled, {"labeled_weak"}), "unlabeled": (unlabeled, {"unlabeled_weak", "unlabeled_strong"})}
    out = {}
    for prefix, (batch, weak_triggers) in views.items():
        out[prefix + "_weak"] = _weak_view(batch) if batch is not None and wanted & weak_triggers else None
        out[prefix + "_strong"] = batch if prefix + "_strong" in wanted else None
    return out["labeled_weak"], out["labeled_strong"], out["unlabeled_weak"], out["unlabeled_strong"]


class WeakStrongDataloader:
    def __init__(self, labeled_loader, unlabeled_loader, batch_contents=("labeled_weak", "labeled_strong", "unlabeled_strong")):
        self.labeled_loader, self.unlabeled_loader = labeled_loader, unlabeled_loader
        self.batch_contents = batch_contents

    def __iter__(self):
        li = iter(self.labeled_loader) if self.labeled_loader is not None else None
        ui = iter(self.unlabeled_loader) if self.unlabeled_loader is not None else None
        while True:
            yield unpack_data_weak_strong(next(li) if li is not None else None, next(ui) if ui is not None else None,
                                          batch_contents=self.batch_contents)


class SyntheticDetectionLoader:
    """Infinite stream of synthetic COCO-style dicts ({"image": strong view, "img_weak": weak view, "instances"})."""
    def __init__(self, batch_size, h, w, num_classes, seed, labeled: bool, boxes_per_image=(5, 20), fixed: bool = False):
        self.bs, self.h, self.w, self.K, self.seed, self.labeled = batch_size, h, w, num_classes, seed, labeled
        self.boxes_per_image, self.fixed = boxes_per_image, fixed

    def __iter__(self):
        it = 0
        while True:
            g = torch.Generator().manual_seed(self.seed + (0 if self.fixed else it))
            batch = []
            for _ in range(self.bs):
                nb = int(torch.randint(self.boxes_per_image[0], self.boxes_per_image[1] + 1, (1,), generator=g))
                img, inst = synthetic.make_image(self.h, self.w, nb, self.K, g)
                if not self.labeled:
                    inst = {"image_size": (self.h, self.w), "gt_boxes": torch.zeros(0, 4), "gt_classes": torch.zeros(0, dtype=torch.int64)}
                batch.append({"image": synthetic.strong_view(img, g), WEAK_IMG_KEY: img, "instances": inst})
            it += 1
            yield batch
