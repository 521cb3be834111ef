"""Drop-in for metrabs_pytorch/multiperson/person_detector.py:PersonDetector.

The reference class owns an ultralytics YOLOv8 network and wraps it with a gamma-correct resize /
pad in front and a box rescale behind (person_detector.py:14-54).  The network is third party and
out of scope; here it is injected, and the two wrappers run as HIP kernels (K9,
csrc/detector_pre.hip) -- same arguments, same return value:

    detector = PersonDetector(ultralytics.YOLO('yolov8m.pt'))
    boxes = detector(images_u8, threshold, nms_iou_threshold, max_detections)   # list of [n_i, 5]

`network` is anything with ultralytics' `predict(source=..., conf=..., iou=..., max_det=...,
classes=[0], verbose=False)` returning objects with `.boxes.xyxy` / `.boxes.conf`, or a plain
callable `network(images_f32, threshold, nms_iou_threshold, max_detections)` returning one
[n_i, 5] tensor (x1, y1, x2, y2, conf) per image in the padded network frame.
"""
import torch

from metrabs_amd import kernels


class PersonDetector(torch.nn.Module):
    def __init__(self, network, input_size=416, autocast_dtype=torch.float16):
        super().__init__()
        self.input_size = input_size  # person_detector.py:12
        self.network = network
        self.autocast_dtype = autocast_dtype  # person_detector.py:35

    def preprocess(self, images):
        """person_detector.py:15-33 -> (network input [N,3,out_h,out_w] f32, geometry)."""
        return kernels.detector_preprocess(images, input_size=self.input_size)

    def _run_network(self, x, threshold, nms_iou_threshold, max_detections):
        if hasattr(self.network, 'predict'):
            with torch.autocast(dtype=self.autocast_dtype, device_type='cuda',
                                enabled=self.autocast_dtype is not None):
                results = self.network.predict(
                    source=x, conf=threshold, iou=nms_iou_threshold, max_det=max_detections,
                    classes=[0], verbose=False)
            return [torch.cat([r.boxes.xyxy.float(), r.boxes.conf.float()[:, None]], dim=1)
                    for r in results]
        return list(self.network(x, threshold, nms_iou_threshold, max_detections))

    def forward(self, images, threshold, nms_iou_threshold, max_detections):
        images = torch.as_tensor(images)
        if not images.is_cuda:
            images = images.cuda()
        x, geom = self.preprocess(images)
        per_image = self._run_network(x, threshold, nms_iou_threshold, max_detections)
        n_per_image = [len(b) for b in per_image]
        if sum(n_per_image) == 0:
            return [torch.zeros(0, 5, device=images.device) for _ in per_image]
        flat = torch.cat([b.to(images.device).float().reshape(-1, 5) for b in per_image])
        return list(torch.split(kernels.detector_scale_boxes(flat, geom), n_per_image))
