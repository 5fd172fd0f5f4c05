module BatchMain (renderDirectory) where

-- The Haskell side of the multi-GPU batch path: what the directory branch of doStart (app/Main.hs:68-77) becomes.
-- Written against the reference's own modules (ConfigFile) and RaytracerFFI.hs beside it; NOT compiled where it was written
-- (no GHC in that image) -- run.sh type-checks it together with the shim wherever the pinning kit is run.

import Control.Monad (forM, filterM, when)
import Data.Yaml (decodeFileEither, prettyPrintParseException)
import System.Directory (doesFileExist)
import System.FilePath (takeBaseName, (</>), (<.>))
import System.IO (hFlush, stdout)

import ConfigFile
import RaytracerFFI

-- Util.promptOverwriteFile (src/Util.hs:21-31) as a question asked BEFORE the batch is handed over: the library writes the files itself.
mayWrite :: FilePath -> IO Bool
mayWrite path = do
  exists <- doesFileExist path
  if not exists then return True else do
    putStr $ "Overwrite " ++ path ++ "? [y/N] "
    hFlush stdout
    answer <- getLine
    let yes = answer == "y" || answer == "Y"
    when (not yes) $ putStrLn "Nothing was written."
    return yes

-- Every scene file of a directory on every GPU of the node: decode them all first (handleScene's messages, app/Main.hs:83-91),
-- apply prepareScene (app/Main.hs:93-103; passed in because it lives in module Main), settle the overwrite questions, then ONE
-- foreign call renders, blooms, encodes and writes scene i on GPU i mod N (renderScenesToFiles -> bs_render_png_files).
-- A scene that fails to decode is reported and skipped, like handleScene does.
renderDirectory :: Bool -> Bool -> (Config -> Bool -> Config) -> [GpuTree] -> FilePath -> [FilePath] -> IO ()
renderDirectory pvw forceWrite prepare gpus outdir inputFiles = do
  decoded <- forM inputFiles $ \f -> do
    putStrLn $ "Reading " ++ f ++ "..."
    r <- decodeFileEither f
    case r of
      Left err  -> putStrLn (prettyPrintParseException err) >> return []
      Right cfg -> do
        putStrLn "Scene successfully read."
        let name = (if pvw then "prev-" else "") ++ takeBaseName f
        return [(prepare cfg pvw, outdir </> name <.> ".png")]
  jobs <- if forceWrite then return (concat decoded) else filterM (mayWrite . snd) (concat decoded)
  putStrLn $ "Rendering " ++ show (length jobs) ++ " scenes on " ++ show (length gpus) ++ " GPU(s)..."
  renderScenesToFiles gpus jobs
  putStrLn "Everything done. Thank you!"
