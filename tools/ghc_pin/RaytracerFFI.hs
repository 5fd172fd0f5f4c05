{-# LANGUAGE ForeignFunctionInterface #-}
module RaytracerFFI (GpuTree, withGpuTree, renderGpu, renderPure, renderToFile,
                     withGpuTrees, renderScenesToFiles, renderBatch) where

import Foreign
import Foreign.C.Types
import Foreign.C.String (CString, peekCString, withCString)
import qualified Data.ByteString as B
import Control.Exception (bracket)
import Control.Concurrent (runInBoundThread)
import Control.Concurrent.MVar (MVar, newMVar, withMVar)
import Control.Monad (when, forM, forM_)
import Data.IORef
import System.IO.Unsafe (unsafePerformIO)
import qualified Data.KdMap.Static as K
import Control.DeepSeq (deepseq)
import Data.List (partition)
import Data.Massiv.Array (U)                                     -- massiv re-uses Prelude / Control.Monad names (map, zip, forM_ ...): the rest
import qualified Data.Massiv.Array as A                          -- stays qualified, as src/Raytracer.hs:11,14 does with `A.map` / `Prelude as P`
import Data.Massiv.Array.IO (Image)                             -- type Image r cs e = Array r Ix2 (Pixel cs e), as src/Raytracer.hs uses it
import Data.Massiv.Array.Manifest.Vector (fromVector)          -- the function src/ImageFilters.hs:78 itself uses (massiv 0.2.x)
import qualified Data.Vector.Storable as VS
import Graphics.ColorSpace (Pixel(..), RGB)
import Linear (V3(..))
import ConfigFile
import StarMap (StarTree)

data BsCtx
-- the context plus its page-locked image buffers, which every frame reuses (allocating page-locked memory costs milliseconds;
-- filling it does not fault): [(buffer, capacity in doubles)] -- one for renderGpu, two for renderBatch (two frames in flight per GPU)
-- gpuLock: a bs_ctx is driven by one thread at a time (include/blackstar_gpu.h); every single-context entry point below takes it, so two
-- thunks of renderPure sparked in parallel queue up instead of racing on the context and its one reusable buffer; the multi-context
-- calls hold every context they were given (withAllLocks).
data GpuTree = GpuTree { gpuCtx :: Ptr BsCtx, gpuBufs :: IORef [(ForeignPtr CDouble, Int)], gpuLock :: MVar () }

-- include/blackstar_gpu.h
foreign import ccall safe   "bs_create"     c_bs_create  :: CInt -> Ptr () -> CSize -> IO (Ptr BsCtx)
foreign import ccall safe   "bs_render"     c_bs_render  :: Ptr BsCtx -> Ptr () -> Ptr CDouble -> CSize -> IO CInt
foreign import ccall safe   "bs_destroy"    c_bs_destroy :: Ptr BsCtx -> IO ()
foreign import ccall unsafe "bs_last_error" c_bs_error   :: IO (Ptr CChar)
-- page-locked image buffers (no first-touch page faults, the copy engine writes them directly): the default below
foreign import ccall safe   "bs_host_alloc"  c_bs_host_alloc :: Ptr BsCtx -> CSize -> IO (Ptr CDouble)
foreign import ccall unsafe "&bs_host_free"  p_bs_host_free  :: FunPtr (Ptr CDouble -> IO ())
-- doRender to the end on the device (app/Main.hs:105-123): render, bloom, writeImg's pixel map AND its PNG encoder
foreign import ccall safe   "bs_png_bound"  c_bs_png_bound  :: CInt -> CInt -> Ptr CSize -> IO CInt
foreign import ccall safe   "bs_render_png" c_bs_render_png :: Ptr BsCtx -> Ptr () -> CDouble -> CInt -> Ptr Word8 -> CSize -> Ptr CSize -> IO CInt
-- batch mode (app/Main.hs:68-77) over every GPU of the node: one context per device, frame i on context i mod N, ONE foreign call
foreign import ccall safe   "bs_device_count" c_bs_device_count :: IO CInt     -- (safe: the first HIP call of the process initialises the runtime)
foreign import ccall safe   "bs_render_png_files" c_bs_render_png_files
  :: Ptr (Ptr BsCtx) -> CInt -> Ptr () -> CInt -> Ptr CDouble -> Ptr CInt -> Ptr CString -> CInt -> IO CInt
foreign import ccall safe   "bs_render_batch" c_bs_render_batch
  :: Ptr (Ptr BsCtx) -> CInt -> Ptr () -> CInt -> Ptr (Ptr CDouble) -> IO CInt
foreign import ccall unsafe "bs_abi_version" c_bs_abi_version :: IO CInt        -- must be 5 (BS_ABI_VERSION this shim was written against)

-- struct bs_star  { double x,y,z,hue,sat; int32 mag; int32 _pad; }   = 48 bytes
pokeStar :: Ptr () -> Int -> (V3 Double, (Int, Double, Double)) -> IO ()
pokeStar base i (V3 x y z, (mag, hue, sat)) = do
  let p = base `plusPtr` (48 * i)
  forM_ (zip [0 ..] [x, y, z, hue, sat]) $ \(k, v) -> pokeByteOff p (8 * k) (realToFrac v :: CDouble)
  pokeByteOff p 40 (fromIntegral mag :: Int32)
  pokeByteOff p 44 (0 :: Int32)

-- The star set as the bs_star array bs_create copies (marshalled ONCE, however many devices get a context).
withStars :: StarTree -> (Ptr () -> Int -> IO a) -> IO a
withStars tree act = do
  let stars = K.assocs tree
      n     = length stars
  v <- c_bs_abi_version
  when (v /= 5) $ ioError (userError ("libblackstar_gpu has ABI version " ++ show v ++ ", this shim expects 5"))
  allocaBytes (48 * max 1 n) $ \buf -> do
    forM_ (zip [0 ..] stars) $ \(i, s) -> pokeStar buf i s
    act buf n

lastError :: String -> IO a
lastError what = c_bs_error >>= peekCString >>= \e -> ioError (userError (what ++ ": " ++ e))

-- bs_last_error() is thread-local in the library, and under -threaded a `safe` foreign call may run on another OS thread than the next
-- call of the same Haskell thread: a call that can fail and the fetch of its message are made from ONE bound thread.
checked :: String -> IO CInt -> IO ()
checked what call = runInBoundThread $ do
  rc <- call
  when (rc /= 0) $ lastError what

checkedPtr :: String -> IO (Ptr a) -> IO (Ptr a)
checkedPtr what call = runInBoundThread $ do
  p <- call
  when (p == nullPtr) $ lastError what
  return p

-- One context on one device; destroyed when the action returns or throws.
withCtx :: Ptr () -> Int -> Int -> (GpuTree -> IO a) -> IO a
withCtx buf n device act =
  bracket (checkedPtr ("bs_create on device " ++ show device) (c_bs_create (fromIntegral device) buf (fromIntegral n))) c_bs_destroy $ \ctx -> do
    ref <- newIORef []
    lock <- newMVar ()
    act (GpuTree ctx ref lock)

-- Upload the star set once (replaces handing `tree` to doStart, app/Main.hs:46-49).
withGpuTree :: Int -> StarTree -> (GpuTree -> IO a) -> IO a
withGpuTree device tree act = withStars tree $ \buf n -> withCtx buf n device act

-- The same on several devices: one context (its own copy of the star index, 22.6 MB) per entry of `devices`; [] = every device
-- bs_device_count reports.  The contexts are destroyed in reverse order when the action returns or throws.
withGpuTrees :: [Int] -> StarTree -> ([GpuTree] -> IO a) -> IO a
withGpuTrees devices tree act = do
  devs <- if not (null devices) then return devices else do
    c <- c_bs_device_count
    when (c <= 0) $ lastError "bs_device_count"
    return [0 .. fromIntegral c - 1]
  withStars tree $ \buf n -> withMany (withCtx buf n) devs act

-- Every context of a multi-context call is held for the length of the call, taken in list order (two batch calls over the same
-- contexts queue up; the same list order everywhere rules out a deadlock between them).
withAllLocks :: [GpuTree] -> IO a -> IO a
withAllLocks gpus act = foldr (\g inner -> withMVar (gpuLock g) (\_ -> inner)) act gpus

-- k page-locked image buffers of at least n doubles each, grown on demand and then reused by every frame (bs_host_free runs when
-- the GC drops a buffer; the memory is hipHostMallocPortable: every device of the node may write it, and it may outlive the context).
imageBuffers :: GpuTree -> Int -> Int -> IO [ForeignPtr CDouble]
imageBuffers (GpuTree ctx ref _) k n = do
  have <- readIORef ref
  let (fit, _tooSmall) = partition ((>= n) . snd) have            -- (buffers that are too small are dropped: the GC frees them)
      keep = take k fit
  fresh <- forM [1 .. k - length keep] $ \_ -> do
    p <- checkedPtr "bs_host_alloc" (c_bs_host_alloc ctx (fromIntegral (8 * n)))
    fp <- newForeignPtr p_bs_host_free p
    return (fp, n)
  writeIORef ref (keep ++ fresh ++ drop k fit)
  return (map fst (keep ++ fresh))

-- struct bs_config { double cam_pos[3], cam_lookat[3], cam_up[3], fov, step_size, star_intensity,
--                    star_saturation, disk_hsi[3], disk_opacity, disk_inner, disk_outer;
--                    int32 width, height, supersampling, _pad; }   = 19 doubles + 4 int32 = 168 bytes
pokeConfig :: Ptr () -> Config -> IO ()
pokeConfig p cfg = do
  let scn = scene cfg; cam = camera cfg
      V3 px py pz = position cam; V3 lx ly lz = lookAt cam; V3 ux uy uz = upVec cam
      PixelHSI dh ds di = diskColor scn          -- hue already /360 (src/ConfigFile.hs:51)
      (w, h) = resolution scn
      ds' = [px,py,pz, lx,ly,lz, ux,uy,uz, fov cam, stepSize scn, starIntensity scn, starSaturation scn,
             dh,ds,di, diskOpacity scn, diskInner scn, diskOuter scn]   -- AS PARSED: radii un-squared
  forM_ (zip [0 ..] ds') $ \(k, v) -> pokeByteOff p (8 * k) (realToFrac v :: CDouble)
  forM_ (zip [0 ..] [w, h, fromEnum (supersampling scn), 0]) $ \(k, v) -> pokeByteOff p (152 + 4 * k) (fromIntegral v :: Int32)

-- Drop-in for `render cfg tree` (src/Raytracer.hs:53): same Config, same result type.
renderGpu :: GpuTree -> Config -> IO (Image U RGB Double)
renderGpu gpu@(GpuTree ctx _ lock) cfg = withMVar lock $ \_ -> do
  let (w, h) = resolution (scene cfg)
      n = w * h * 3
  -- the caller owns the image: one page-locked buffer (bs_host_alloc), reused frame after frame.
  -- (mallocForeignPtrArray n also works, but a fresh pageable buffer per frame costs 8.7 instead of 4.5 ms per 1080p frame:
  --  bench.py's "boundary" block, bs_render_pageable vs bs_render_pinned.)
  [fp] <- imageBuffers gpu 1 n
  checked "bs_render" $ allocaBytes 168 $ \pc -> pokeConfig pc cfg >> withForeignPtr fp (\po -> c_bs_render ctx pc po (fromIntegral n))
  -- interleaved RGB f64, row-major, y down == the Storable layout of `Pixel RGB Double`: view the buffer as a storable vector
  -- (no copy) and let massiv convert it to its unboxed planar form -- a COPY, so the buffer is free for the next frame.
  -- (fromVector is what the reference's own boxBlur returns its result with; newer massiv also has
  --  Data.Massiv.Array.Unsafe.unsafeArrayFromForeignPtr0 + computeIO for the same purpose.)
  unboxedCopy fp w h

-- `render cfg tree` as the PURE value app/Main.hs:109 hands to timeAction (src/Util.hs:37-45 forces it with deepseq and times that):
-- img <- timeAction "Rendering" $ renderPure gpu cfg.  The thunk MUST be forced inside the withGpuTree bracket that made `gpu` (timeAction
-- does, at once): forced later it would call bs_render on a destroyed context.  Frames of one context are serialised by gpuLock.
renderPure :: GpuTree -> Config -> Image U RGB Double
renderPure gpu cfg = unsafePerformIO (renderGpu gpu cfg)
{-# NOINLINE renderPure #-}

-- Copy a frame out of its (reusable) page-locked buffer into massiv's unboxed planar form.
unboxedCopy :: ForeignPtr CDouble -> Int -> Int -> IO (Image U RGB Double)
unboxedCopy fp w h = do
  let vec = VS.unsafeFromForeignPtr0 (castForeignPtr fp) (w * h) :: VS.Vector (Pixel RGB Double)
      img = fromVector A.Par (h A.:. w) vec :: Image U RGB Double
  img `deepseq` touchForeignPtr fp                                 -- the copy has happened before the buffer can be reused
  return img

-- Replaces the tail of doRender (app/Main.hs:109-123): img <- render cfg tree; final <- bloom ... img; writeImg outPath final.
-- The file's bytes are made on the GPU; they decode to what writeImg's `A.map (toWord8 . fmap sRGB)` feeds its encoder.
renderToFile :: GpuTree -> Config -> FilePath -> IO ()
renderToFile (GpuTree ctx _ lock) cfg outPath = withMVar lock $ \_ -> do
  let scn = scene cfg
      (w, h) = resolution scn
  cap <- alloca $ \p -> do
    checked "bs_png_bound" $ c_bs_png_bound (fromIntegral w) (fromIntegral h) p
    peek p
  allocaBytes (fromIntegral cap) $ \buf -> alloca $ \pn -> do      -- (a reused bs_host_alloc buffer is written by the GPU itself)
    checked "bs_render_png" $ allocaBytes 168 $ \pc -> pokeConfig pc cfg >>
            c_bs_render_png ctx pc (realToFrac (bloomStrength scn)) (fromIntegral (bloomDivider scn)) buf cap pn
    n <- peek pn
    B.packCStringLen (castPtr buf, fromIntegral n) >>= B.writeFile outPath

-- Replaces the directory loop of doStart (app/Main.hs:68-77: forM_ ... handleScene, i.e. doRender per scene file) in ONE foreign call:
-- scene i is rendered, bloomed (strength 0 = no bloom, like app/Main.hs:113) and PNG-encoded on gpus !! (i mod N) and written to its
-- path while later scenes render -- per GPU a rolling pipeline of frames in flight, its own file buffers and its own writer thread on the
-- GPU's NUMA node; no collective, no GPU waits for another, only the file's bytes cross PCIe.  Files are created
-- or truncated (what --force does; ask promptOverwriteFile BEFORE the call for the others, see the doStart edit below).
renderScenesToFiles :: [GpuTree] -> [(Config, FilePath)] -> IO ()
renderScenesToFiles gpus jobs = do
  let n    = length jobs
      scns = map (scene . fst) jobs
  when (null gpus) $ ioError (userError "renderScenesToFiles: no GPU contexts")
  when (n > 0) $ withAllLocks gpus $
    withArray (map gpuCtx gpus) $ \pctxs ->
    allocaBytes (168 * n) $ \pcfgs ->
    withArray (map (realToFrac . bloomStrength) scns :: [CDouble]) $ \pstrengths ->
    withArray (map (fromIntegral . bloomDivider) scns :: [CInt]) $ \pdividers ->
    withMany withCString (map snd jobs) $ \cpaths ->
    withArray cpaths $ \ppaths -> do
      forM_ (zip [0 ..] jobs) $ \(i, (cfg, _)) -> pokeConfig (pcfgs `plusPtr` (168 * i)) cfg
      checked "bs_render_png_files" $
        c_bs_render_png_files pctxs (fromIntegral (length gpus)) pcfgs (fromIntegral n) pstrengths pdividers ppaths 0

-- For a caller that keeps Haskell's bloom / writeImg: the frames of `cfgs` through bs_render_batch, frame i on gpus !! (i mod N), handed
-- to `consume i img` in order.  Rounds of 2N frames go into 2N page-locked buffers (two per context, reused by every round; the kernels
-- write them directly), so page-locked memory stays at 2N frames however long the animation is; each frame is copied into massiv's
-- unboxed form before its buffer is reused.  Every round starts at a multiple of N, so frame i does run on context i mod N.
renderBatch :: [GpuTree] -> [Config] -> (Int -> Image U RGB Double -> IO ()) -> IO ()
renderBatch gpus cfgs consume = withAllLocks gpus $ do
  when (null gpus) $ ioError (userError "renderBatch: no GPU contexts")
  let nGpu   = length gpus
      sizeOf' cfg = let (w, h) = resolution (scene cfg) in w * h * 3
      largest = maximum (1 : map sizeOf' cfgs)
      rounds _ [] = []
      rounds i xs = let (a, b) = splitAt (2 * nGpu) xs in (i, a) : rounds (i + length a) b
  perGpu <- forM gpus $ \g -> imageBuffers g 2 largest              -- [[first, second]] per context
  let slots = map head perGpu ++ map (head . tail) perGpu           -- slot j of a round belongs to context j mod N
  withArray (map gpuCtx gpus) $ \pctxs ->
    forM_ (rounds 0 cfgs) $ \(first, chunk) -> do
      let m = length chunk
          bufs = take m slots
      allocaBytes (168 * m) $ \pcfgs -> do
        forM_ (zip [0 ..] chunk) $ \(j, cfg) -> pokeConfig (pcfgs `plusPtr` (168 * j)) cfg
        checked "bs_render_batch" $ withMany withForeignPtr bufs $ \ptrs -> withArray ptrs $ \pouts ->
                c_bs_render_batch pctxs (fromIntegral nGpu) pcfgs (fromIntegral m) pouts
      forM_ (zip3 [first ..] chunk bufs) $ \(i, cfg, fp) -> do
        let (w, h) = resolution (scene cfg)
        unboxedCopy fp w h >>= consume i
