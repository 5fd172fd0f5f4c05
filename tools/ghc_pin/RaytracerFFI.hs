{-# LANGUAGE ForeignFunctionInterface #-}
module RaytracerFFI (GpuTree, withGpuTree, renderGpu, renderToFile) where

import Foreign
import Foreign.C.Types
import Foreign.C.String (peekCString)
import qualified Data.ByteString as B
import Control.Exception (bracket)
import Control.Monad (when, forM_)
import Data.IORef
import qualified Data.KdMap.Static as K
import Control.DeepSeq (deepseq)
import Data.Massiv.Array as A
import Data.Massiv.Array.IO (Image)                             -- type Image r cs e = Array r Ix2 (Pixel cs e), as src/Raytracer.hs uses it
import Data.Massiv.Array.Manifest.Vector (fromVector)          -- the function src/ImageFilters.hs:78 itself uses (massiv 0.2.x)
import qualified Data.Vector.Storable as VS
import Graphics.ColorSpace
import Linear (V3(..))
import ConfigFile
import StarMap (StarTree)

data BsCtx
-- the context plus ONE page-locked image buffer that every frame reuses (allocating page-locked memory costs milliseconds;
-- filling it does not fault): (buffer, capacity in doubles)
data GpuTree = GpuTree { gpuCtx :: Ptr BsCtx, gpuBuf :: IORef (ForeignPtr CDouble, Int) }

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
foreign import ccall unsafe "bs_device_count" c_bs_device_count :: IO CInt     -- one context per device for batch mode
foreign import ccall unsafe "bs_abi_version" c_bs_abi_version :: IO CInt        -- must be 4 (BS_ABI_VERSION this shim was written against)

-- struct bs_star  { double x,y,z,hue,sat; int32 mag; int32 _pad; }   = 48 bytes
pokeStar :: Ptr () -> Int -> (V3 Double, (Int, Double, Double)) -> IO ()
pokeStar base i (V3 x y z, (mag, hue, sat)) = do
  let p = base `plusPtr` (48 * i)
  forM_ (zip [0 ..] [x, y, z, hue, sat]) $ \(k, v) -> pokeByteOff p (8 * k) (realToFrac v :: CDouble)
  pokeByteOff p 40 (fromIntegral mag :: Int32)
  pokeByteOff p 44 (0 :: Int32)

-- Upload the star set once (replaces handing `tree` to doStart, app/Main.hs:46-49).
withGpuTree :: Int -> StarTree -> (GpuTree -> IO a) -> IO a
withGpuTree device tree act = do
  let stars = K.assocs tree
      n     = length stars
  allocaBytes (48 * max 1 n) $ \buf -> do
    forM_ (zip [0 ..] stars) $ \(i, s) -> pokeStar buf i s
    v <- c_bs_abi_version
    when (v /= 4) $ ioError (userError ("libblackstar_gpu has ABI version " ++ show v ++ ", this shim expects 4"))
    bracket (c_bs_create (fromIntegral device) buf (fromIntegral n)) c_bs_destroy $ \ctx -> do
      when (ctx == nullPtr) $ c_bs_error >>= peekCString >>= \e -> ioError (userError ("bs_create: " ++ e))
      nullBuf <- newForeignPtr_ nullPtr
      ref <- newIORef (nullBuf, 0)
      act (GpuTree ctx ref)

-- The page-locked image buffer, grown on demand and then reused by every frame (bs_host_free runs when the GC drops it).
imageBuffer :: GpuTree -> Int -> IO (ForeignPtr CDouble)
imageBuffer (GpuTree ctx ref) n = do
  (fp, cap) <- readIORef ref
  if cap >= n then return fp else do
    p <- c_bs_host_alloc ctx (fromIntegral (8 * n))
    when (p == nullPtr) $ c_bs_error >>= peekCString >>= \e -> ioError (userError ("bs_host_alloc: " ++ e))
    fp' <- newForeignPtr p_bs_host_free p
    writeIORef ref (fp', n)
    return fp'

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
renderGpu gpu@(GpuTree ctx _) cfg = do
  let (w, h) = resolution (scene cfg)
      n = w * h * 3
  -- the caller owns the image: one page-locked buffer (bs_host_alloc), reused frame after frame.
  -- (mallocForeignPtrArray n also works, but a fresh pageable buffer per frame costs 8.7 instead of 4.5 ms per 1080p frame:
  --  bench.py's "boundary" block, bs_render_pageable vs bs_render_pinned.)
  fp <- imageBuffer gpu n
  rc <- allocaBytes 168 $ \pc -> pokeConfig pc cfg >> withForeignPtr fp (\po -> c_bs_render ctx pc po (fromIntegral n))
  when (rc /= 0) $ c_bs_error >>= peekCString >>= \e -> ioError (userError ("bs_render: " ++ e))
  -- interleaved RGB f64, row-major, y down == the Storable layout of `Pixel RGB Double`: view the buffer as a storable vector
  -- (no copy) and let massiv convert it to its unboxed planar form -- a COPY, so the buffer is free for the next frame.
  -- (fromVector is what the reference's own boxBlur returns its result with; newer massiv also has
  --  Data.Massiv.Array.Unsafe.unsafeArrayFromForeignPtr0 + computeIO for the same purpose.)
  let vec = VS.unsafeFromForeignPtr0 (castForeignPtr fp) (w * h) :: VS.Vector (Pixel RGB Double)
      img = fromVector Par (h :. w) vec :: Image U RGB Double
  img `deepseq` touchForeignPtr fp                                 -- the copy has happened before the buffer can be reused
  return img

-- Replaces the tail of doRender (app/Main.hs:109-123): img <- render cfg tree; final <- bloom ... img; writeImg outPath final.
-- The file's bytes are made on the GPU; they decode to what writeImg's `A.map (toWord8 . fmap sRGB)` feeds its encoder.
renderToFile :: GpuTree -> Config -> FilePath -> IO ()
renderToFile (GpuTree ctx _) cfg outPath = do
  let scn = scene cfg
      (w, h) = resolution scn
      failWith what = c_bs_error >>= peekCString >>= \e -> ioError (userError (what ++ ": " ++ e))
  cap <- alloca $ \p -> do
    rc <- c_bs_png_bound (fromIntegral w) (fromIntegral h) p
    when (rc /= 0) $ failWith "bs_png_bound"
    peek p
  allocaBytes (fromIntegral cap) $ \buf -> alloca $ \pn -> do      -- (a reused bs_host_alloc buffer is written by the GPU itself)
    rc <- allocaBytes 168 $ \pc -> pokeConfig pc cfg >>
            c_bs_render_png ctx pc (realToFrac (bloomStrength scn)) (fromIntegral (bloomDivider scn)) buf cap pn
    when (rc /= 0) $ failWith "bs_render_png"
    n <- peek pn
    B.packCStringLen (castPtr buf, fromIntegral n) >>= B.writeFile outPath
