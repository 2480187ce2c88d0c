// TypeScript surface of the Node host (tsc is not part of the build image; this file is the typed contract).
/// <reference types="node" />
export type Direction = 'vertical' | 'horizontal';
export type StitchMode = 'min' | 'max' | 'original';
export interface StitchImage {
  width: number;            // naturalWidth  (pages/index/index.js:724-739)
  height: number;           // naturalHeight
  data?: Uint8Array;        // RGBA8, straight alpha, row-major, width*height*4 bytes
  orientation?: 1 | 2 | 3 | 4 | 5 | 6 | 7 | 8;
  fileSize?: number;        // bytes, feeds bigTask (index.js:1211-1212)
  opaque?: boolean;         // hint: all alpha bytes are 255
  bmpWidth?: number; bmpHeight?: number;
}
export interface StitchOptions {
  mode?: StitchMode; gap?: number; filter?: 'bilinear' | 'nearest' | 'area';      // 'area': box-average minified axes (one reading of imageSmoothingQuality 'high'); default bilinear
  platform?: 'ios' | 'android' | 'devtools' | 'windows' | 'mac' | 'other';
  maxSide?: number; maxPixels?: number; superSample?: number;
  edgeAA?: boolean;                         // anti-alias fractional rectangle edges (ctx.scale(superSample), unrounded cursor); default: true iff `platform` is given
  onProgress?: (percent: number) => void;   // stitchProgress checkpoints (index.js:1193-1611)
  pngLevel?: 0 | 1;                         // PNG export form: 0 stored, 1 compressed on the GPU (process-wide once set)
  devices?: number[];                       // GPUs to shard the stitch over from this process; devices[0] is the root (RCCL gather over xGMI)
  split?: 'image' | 'band' | 'rows' | 'auto';   // with devices: image i -> devices[i mod n] | equal output rows per GPU, draw by draw | GPU s owns canvas rows across all draws (horizontal strips: full-width bands) | default 'auto': 'image' when its parts are full-width, else 'rows'
}
export interface PlanRect { image: number; orientation: number; dx: number; dy: number; dw: number; dh: number; }
export interface StitchPlan {
  outW: number; outH: number; scaleDown: number; superSample: number; canvasW: number; canvasH: number;
  bigTask: boolean; rects: PlanRect[];
}
export interface StitchResult { width: number; height: number; data: Buffer; plan: StitchPlan; }
export function stitch(images: StitchImage[], direction: Direction, opts?: StitchOptions): Promise<StitchResult | null>;
export function stitchSync(images: StitchImage[], direction: Direction, opts?: StitchOptions): StitchResult | null;
export function plan(images: StitchImage[], direction: Direction, opts?: StitchOptions): StitchPlan | null;
export interface StitchPngResult { width: number; height: number; png: Buffer; plan: StitchPlan; }
export function stitchPng(images: StitchImage[], direction: Direction, opts?: StitchOptions): Promise<StitchPngResult | null>;
export function encodePng(data: Uint8Array, width: number, height: number, opts?: { pngLevel?: 0 | 1 }): Buffer;
export function setPngLevel(level: 0 | 1): void;
export function decodePng(file: Uint8Array): { width: number; height: number; data: Buffer };
export function stitchFiles(paths: string[], direction: Direction, opts?: StitchOptions, outPath?: string): Promise<StitchPngResult | null>;
export function decodeImage(file: Uint8Array): { width: number; height: number; orientation: number; opaque: boolean; data: Buffer };
