// Types of the nodencl-shaped surface implemented by index.js (what phaneron imports from 'nodencl').
/// <reference types="node" />

export interface ImageDims {
	width: number
	height: number
}

export interface RunTimings {
	dataToKernel: number
	kernelExec: number
	totalTime: number
}

export type BufDir = 'readonly' | 'writeonly' | 'readwrite'
export type BufSVMType = 'none' | 'coarse' | 'fine'
export type HostAccessDir = 'readonly' | 'writeonly' | 'none'

export interface OpenCLBuffer extends Buffer {
	readonly numBytes: number
	readonly owner: string
	readonly imageDims?: ImageDims
	readonly creationTime: [number, number]
	timestamp: number
	loadstamp: number
	hostAccess(dir: HostAccessDir, queue?: number, src?: Buffer): Promise<void>
	addRef(): void
	release(): void
	refCount(): number
	/** staging extension: device -> mirror on `queue` (default unload) without a host wait */
	downloadAsync(queue?: number): void
}

/** staging extension: a recorded point in a queue */
export interface QueueEvent {
	wait(): Promise<void>
	done(): boolean
}

export interface RouteLink {
	readonly rank: number
	readonly world: number
	/** sends and receives of one frame period go inside one group */
	group(fn: () => void): void
	send(buf: OpenCLBuffer, peer: number): void
	recv(buf: OpenCLBuffer, peer: number): void
	/** the communication stream waits for everything enqueued so far on `queue` */
	afterQueue(queue?: number): void
	/** `queue` waits for everything enqueued so far on the communication stream */
	queueAfter(queue?: number): void
	wait(): void
}

export interface OpenCLProgram {
	readonly name: string
	readonly globalWorkItems: number[]
	readonly workItemsPerGroup: number
}

export interface KernelParams {
	[key: string]: unknown
}

export interface PlatformInfo {
	vendor: string
	name: string
	devices: Array<{ type: string; name: string; vendor: string }>
}

export interface DeferredStats {
	pending: number; recorded: number; launched: number; fused: number; fusedNodes: number; plain: number; dropped: number; fallbacks: number; lastFallback: string | null
	/** frames that went to the device together with other channels' frames, several to a launch */
	batched?: number
}

export interface BufferStats {
	/** buffers somebody owns (parked ones are nobody's) */
	liveBuffers: number; liveBytes: number
	/** device bytes in the library's own pool */
	pooledBytes: number
	/** released frames / images kept whole for the next createBuffer of their shape */
	parkedBuffers: number; parkedBytes: number
	/** pinned host mirrors: attached to buffers, pooled, the most ever attached at once, hipHostMalloc calls so far */
	pinnedInUse: number; pinnedPooled: number; pinnedPeak: number; pins: number
}

export class clContext {
	/** `deferred` (default TRUE since round 4; `false`, or PHANERON_DEFERRED=0 in the environment, gives the launch-as-posted context):
	 * runProgram records instead of launching; a packed frame's recorded operator chain reaches the device as one fused kernel when its
	 * result is asked for (hostAccess 'readonly', downloadAsync, a route send, realise).  RunTimings of recorded jobs are zeros and
	 * waitFinish(queue.process) returns at once - INTEGRATION.md 3a.
	 * `profile` on a deferred context: a frame's terminal `write` is launched where it is posted and returns the fused launch's device time.
	 * `earlyLaunch` (default false; PHANERON_EARLY_LAUNCH=1): launch a frame at the end of the tick that posted its terminal `write`.
	 * `recycleBuffers` (default true; PHANERON_RECYCLE=0): released frames / images are parked for the next createBuffer of their shape,
	 * up to `parkMb` MiB (default 4096) or the most that was ever in use at once.  A parked buffer is taken over as the same JS object;
	 * `strictHandles` (default false; PHANERON_STRICT_HANDLES=1): a fresh object per takeover, so that a reference kept past release()
	 * is refused ('... released buffer') instead of aliasing the next owner's buffer - for running an application under test. */
	constructor(params?: { platformIndex?: number; deviceIndex?: number; overlapping?: boolean; profile?: boolean; spinWaitMicros?: number; deferred?: boolean;
		earlyLaunch?: boolean; recycleBuffers?: boolean; parkMb?: number; strictHandles?: boolean })
	readonly queue: { load: number; process: number; unload: number }
	initialise(): Promise<void>
	getPlatformInfo(): PlatformInfo
	createBuffer(numBytes: number, bufDir: BufDir, bufType: BufSVMType, imageDims?: ImageDims, owner?: string): Promise<OpenCLBuffer>
	createProgram(kernel: string, options: { name: string; globalWorkItems?: number | Uint32Array | number[]; workItemsPerGroup?: number }): Promise<OpenCLProgram>
	runProgram(program: OpenCLProgram, params: KernelParams, queue?: number): Promise<RunTimings>
	/** Recording contexts only: runProgram without the promise (node/jobs.js uses it per job of a batch).  null = not a recording
	 *  context, or a `profile` one - call runProgram; throws what runProgram would reject with. */
	recordProgram(program: OpenCLProgram, params: KernelParams, queue?: number): RunTimings | null
	waitFinish(queue?: number): Promise<void>
	/** deferred contexts: make these buffers' contents real now, as a consumer on the device would need them */
	realise(...bufs: OpenCLBuffer[]): void
	/** deferred contexts: run everything still recorded; returns the recording's counters (null on a plain context) */
	flushDeferred(): DeferredStats | null
	deferredStats(): DeferredStats | null
	/** extension: note the kernels the calls made from here on (this thread) launch; dryRun: choose and check, enqueue nothing */
	traceBegin(dryRun?: boolean): void
	/** extension: the kernels launched since traceBegin, '+'-joined, e.g. "v210_yadif_pair+compose_up_write_v210" */
	traceEnd(): string
	/** wait until everything launched so far on `queue` has finished (on a deferred context waitFinish(queue.process) returns at once) */
	drain(queue?: number): Promise<void>
	/** staging extension: later work on `waiter` starts after everything enqueued so far on `signal` */
	queueWaitQueue(waiter: number, signal: number): void
	/** staging extension: a point in `queue`; wait() polls on the JS thread for up to `spinWaitMicros` before handing the wait to the libuv pool */
	recordEvent(queue?: number): QueueEvent
	/** ROUTE across GPUs: RCCL send / recv on a communication stream of its own, ordered on the device */
	openRoute(id: Buffer, rank: number, world: number): RouteLink
	static routeUniqueId(): Buffer
	/** library options: 'lds_lut' (0 | 1), 'stream_images' (0 cached | 1 streamed | 2 by size), 'stream_threshold_mb', 'host_pool_mb' (pinned mirrors kept for reuse) */
	setOption(name: string, value: number): void
	/** library options also: 'fail_launches' (tests: every launch fails while set) */
	logBuffers(): BufferStats
	bufferStats(): BufferStats
	/** hand the parked buffers back to the library */
	trim(): void
}

/** which precompiled kernel createProgram(kernelSrc, {name}) selects - needs no context and no GPU */
export function resolveProgram(kernelSrc: string, name: string): { kernel: string; format: string | null; how: 'tag' | 'name' | 'text' | 'signature' }

/** the library's host colour maths (the numbers src/process/colourMaths.ts and transform.ts:119-171 produce) */
export const colour: {
	gamma2linearLUT(colSpec: string): Float32Array
	linear2gammaLUT(colSpec: string): Float32Array
	ycbcr2rgbMatrix(colSpec: string, numBits?: number, lumaBlack?: number, lumaWhite?: number, chromaRange?: number): Float32Array
	rgb2ycbcrMatrix(colSpec: string, numBits?: number, lumaBlack?: number, lumaWhite?: number, chromaRange?: number): Float32Array
	rgb2rgbMatrix(srcColSpec: string, dstColSpec: string): Float32Array
	transformMatrix(width: number, height: number, params?: { flipH?: boolean; flipV?: boolean; anchorX?: number; anchorY?: number; scaleX?: number; scaleY?: number; offsetX?: number; offsetY?: number; rotate?: number }): Float32Array
}
export const FORMATS: string[]
export function planeBytes(format: string, width: number, height: number): number[]
